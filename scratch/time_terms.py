import sys, time, torch
sys.path.insert(0, "/root/repo")
from finetune_fair_diffusion_amd import factory
dev = torch.device("cuda", 0)
def run(wi, wf):
    args = factory.default_args(train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                                val_GPU_batch_size=8, size_face=224, img_size_small=224, weight_loss_img=wi, weight_loss_face=wf)
    tr, _ = factory.build_trainer(args, dev, seed=0, regularisers=True, lora_up_std=0.01)
    tokens = factory.synthetic_tokens(13, 49408)
    torch.manual_seed(5991)
    ts = []
    for i in range(4):
        noises = torch.randn([8, 4, 64, 64])
        torch.cuda.synchronize(); t0 = time.perf_counter()
        tr.train_step(tokens, noises.to(dev), 20)
        torch.cuda.synchronize(); ts.append(time.perf_counter() - t0)
    print(f"weight_loss_img={wi} weight_loss_face={wf}: steps {[round(t*1e3) for t in ts]} ms", flush=True)
    del tr
    torch.cuda.empty_cache()
for wi, wf in [(0.0, 0.0), (8.0, 0.0), (8.0, 1.0), (0.0, 0.0)]:
    run(wi, wf)
