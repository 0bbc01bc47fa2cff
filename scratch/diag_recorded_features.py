"""Why does the RECORDED CLIP / DINO forward of the loss phase take 50-90 ms of host time when the same forward without recording takes ~13 ms?
Times both on a warmed-up trainer and prints the cProfile top of the recorded call."""
import cProfile, io, os, pstats, sys, time
ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)
import finetune_fair_diffusion_amd  # noqa: F401
import torch
from finetune_fair_diffusion_amd import factory
dev = torch.device("cuda", 0)
torch.cuda.set_device(0)
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", size_face=224, img_size_small=224, weight_loss_img=8.0, weight_loss_face=1.0)
tr, models = factory.build_trainer(args, dev, cfgs=factory.SD15, seed=0, rank=0, world_size=1, regularisers=True, experiment="exp-1", lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, 49408)
torch.manual_seed(5991)
for i in range(2):
    tr.train_step(tokens, torch.randn(8, 4, 64, 64), 20)
torch.cuda.synchronize()
images = (torch.rand(8, 3, 512, 512, device=dev) * 2 - 1).half()
small, _ = tr.resize_small(images)
torch.cuda.synchronize()


def timed(record, n=5):
    out = []
    for _ in range(n):
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        e = tr.image_features(small, record=record)
        t1 = time.perf_counter()
        torch.cuda.synchronize()
        t2 = time.perf_counter()
        out.append((round(1e3 * (t1 - t0), 1), round(1e3 * (t2 - t0), 1)))
        tr.clip._ctx = None
        tr.dino._ctx = None
        del e
    return out


print("record=False (host enqueue ms, total ms):", timed(False))
print("record=True  (host enqueue ms, total ms):", timed(True))
print("record=False again:", timed(False))
st = torch.cuda.memory_stats()
print("allocator: num_device_alloc", st.get("num_device_alloc"), "num_alloc_retries", st.get("num_alloc_retries"), "reserved GiB", st["reserved_bytes.all.current"] / 2 ** 30)
pr = cProfile.Profile()
pr.enable()
e = tr.image_features(small, record=True)
pr.disable()
torch.cuda.synchronize()
s = io.StringIO()
pstats.Stats(pr, stream=s).sort_stats("cumulative").print_stats(25)
print(s.getvalue()[:6000])
st2 = torch.cuda.memory_stats()
print("allocator after: num_device_alloc", st2.get("num_device_alloc"))
