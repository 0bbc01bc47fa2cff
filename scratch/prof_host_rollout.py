"""Where does the HOST spend its time while enqueueing a rollout?  cProfile of 20 frozen denoising steps (SD-v1.5, B = 8)."""
import cProfile, os, pstats, sys, time
sys.path.insert(0, os.path.dirname(os.path.dirname(os.path.abspath(__file__))))
import torch
from finetune_fair_diffusion_amd import factory

dev = torch.device("cuda:0")
args = factory.default_args(experiment="exp-1", train_unet=True, train_text_encoder=False, rank=4, train_images_per_prompt_GPU=8, train_GPU_batch_size=3,
                            val_GPU_batch_size=8, mixed_precision="fp16", weight_loss_img=0.0, weight_loss_face=0.0)
CFG = factory.TINY if os.environ.get("FD_TINY") else factory.SD15
tr, models = factory.build_trainer(args, dev, cfgs=CFG, seed=0, regularisers=False, lora_up_std=0.01)
tokens = factory.synthetic_tokens(13, CFG["clip"].vocab_size)
enc = tr.encode_pair(tr.eval_te, tokens)
hw = CFG["unet"].sample_size
n8 = torch.randn(8, 4, hw, hw, device=dev)
for which, unet, kw in (("frozen", tr.eval_unet, {}), ("lora+record", tr.unet, dict(keep_inputs=True, record_prompt=True, keep_activations=True))):
    tr.rollout(unet, enc, n8, 20, **kw); torch.cuda.synchronize()
    t0 = time.perf_counter()
    pr = cProfile.Profile(); pr.enable()
    out = tr.rollout(unet, enc, n8, 20, **kw)
    pr.disable()
    t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"== {which}: host enqueue {1e3 * (t1 - t0):.1f} ms (under cProfile), device done after {1e3 * (t2 - t0):.1f} ms")
    pstats.Stats(pr).sort_stats("tottime").print_stats(22)
    del out
    t0 = time.perf_counter(); out = tr.rollout(unet, enc, n8, 20, **kw); t1 = time.perf_counter(); torch.cuda.synchronize(); t2 = time.perf_counter()
    print(f"== {which}: host enqueue {1e3 * (t1 - t0):.1f} ms (no profiler), device done after {1e3 * (t2 - t0):.1f} ms")
    del out
