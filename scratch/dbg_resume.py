import sys, os, json, torch, tempfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_engine_gpu import _train_argv
from finetune_fair_diffusion_amd import train
from finetune_fair_diffusion_amd.factory import TINY
t = tempfile.mkdtemp()
la, lb, lc = [], [], []
full, _ = train.main(_train_argv(t + "/a", 4), cfgs=TINY, log=la.append)
part, _ = train.main(_train_argv(t + "/b", 2), cfgs=TINY, log=lb.append)
sa = torch.load(t + "/a/checkpoints/checkpoint_tmp-2/trainer_state.pth", weights_only=False)
sb = torch.load(t + "/b/checkpoints/checkpoint_tmp-2/trainer_state.pth", weights_only=False)
for k in ("unet", "text_encoder"):
    print(k, "exp_avg diff", float((sa["exp_avg"][k] - sb["exp_avg"][k]).abs().max()), float(sa["exp_avg"][k].abs().max()))
print("torch rng equal", torch.equal(sa["rng"]["torch"], sb["rng"]["torch"]), "py", sa["rng"]["python"] == sb["rng"]["python"])
res, _ = train.main(_train_argv(t + "/c", 4, ["--resume_from_checkpoint", t + "/b/checkpoints/checkpoint_tmp-2"]), cfgs=TINY, log=lc.append)
for x in la: print("A", x)
for x in lb: print("B", x)
for x in lc: print("C", x)
