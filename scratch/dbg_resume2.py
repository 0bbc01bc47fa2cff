import sys, os, json, torch, tempfile
sys.path.insert(0, 'tests'); sys.path.insert(0, '.')
from test_engine_gpu import _train_argv
from finetune_fair_diffusion_amd import train, checkpoint as ck
from finetune_fair_diffusion_amd.cli import parse_args
from finetune_fair_diffusion_amd.factory import TINY, build_trainer
t = tempfile.mkdtemp()
part, _ = train.main(_train_argv(t + "/b", 2), cfgs=TINY, log=print)
args = parse_args(_train_argv(t + "/c", 4), with_extras=True)
train.set_seed(args.seed, True, 0)
C, _ = build_trainer(args, torch.device("cuda", 0), TINY, seed=args.seed)
ck.load_state(C, t + "/b/checkpoints/checkpoint_tmp-2")
for which in ("unet", "te"):
    a, b = getattr(part, which).lora_bank, getattr(C, which).lora_bank
    for buf in ("flat", "ema", "exp_avg", "exp_avg_sq"):
        print(which, buf, float((getattr(a, buf) - getattr(b, buf)).abs().max()))
print(part.opt_step, C.opt_step, part.lr_step, C.lr_step, [e.optimization_step for e in part.ema], [e.optimization_step for e in C.ema])
tok = train.HashTokenizer(TINY["clip"].vocab_size)("a photo of the face of a nurse, a person")
noise = torch.randn(4, 4, 32, 32)
o1 = part.train_step(tok, noise, 3)
o2 = C.train_step(tok, noise, 3)
print("loss", o1["loss_fair"], o2["loss_fair"])
print("images diff", float((o1["images"].float() - o2["images"].float()).abs().max()), "ori", float((o1["images_ori"].float() - o2["images_ori"].float()).abs().max()))
print("probs", o1["probs"][:, 1], o2["probs"][:, 1])
