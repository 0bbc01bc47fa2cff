"""One rank of a two-process run of the PRODUCT's training step: launched by tests/test_two_rank_gpu.py through
``python -m torch.distributed.run --nproc-per-node 2`` with the gloo backend, both ranks sharing cuda:0 (RCCL refuses two ranks on one device;
the collectives of the step -- probability all-gather, OT-plan all-reduce, flat LoRA-gradient all-reduce -- are backend-agnostic
``torch.distributed`` calls).  Mirrors the reference's launch (exp-1-debias-gender/1-main-debias.py:693, :1746-1749, :1805-1837, :1998-2011):
rank k owns images [k*B, (k+1)*B) of the global batch with its own noise.  Writes what the parent test compares to ``<out>/rank<k>.pt``."""
import os
import sys

import torch
import torch.distributed as dist

HERE = os.path.dirname(os.path.abspath(__file__))
sys.path.insert(0, HERE)
sys.path.insert(0, os.path.dirname(HERE))
import util_models as U  # noqa: E402

B_PER_RANK, S, NOISE_SEED = int(os.environ.get("FD_TEST_B_PER_RANK", "3")), 3, 4242


def build(experiment, dev, rank, world):
    from finetune_fair_diffusion_amd.fairness import EXPERIMENT_ATTRS
    from finetune_fair_diffusion_amd.step import FairnessTrainer
    ncls = EXPERIMENT_ATTRS[experiment][0]
    tt = experiment == "exp-1"           # exp-1: both banks (U-Net + text encoder); exp-3 / exp-4: U-Net bank, OT targets (exp-4: 8-logit head, three attributes)
    sds = U.synthetic_sds(train_unet=True, train_te=tt, lora_up_std=0.05, num_classes=ncls)
    pm = U.product_models(sds, dev, train_unet=True, train_te=tt, num_classes=ncls)
    args = U.make_args(train_unet=True, train_text_encoder=tt, uncertainty_threshold=0.7, train_GPU_batch_size=B_PER_RANK)
    tr = FairnessTrainer(args, pm["text_encoder"], pm["unet"], pm["vae"], pm["classifier"], pm["scheduler"], eval_text_encoder=pm["eval_text_encoder"],
                         eval_unet=pm["eval_unet"], experiment=experiment, rank=rank, world_size=world, device=dev)
    return tr


def global_noises(world):
    return torch.randn(world * B_PER_RANK, 4, 32, 32, generator=torch.Generator().manual_seed(NOISE_SEED))


def snapshot(tr, out):
    return dict(targets={k: v.clone() for k, v in out["targets_by_attr"].items()}, loss_fair=out["loss_fair"].clone(), finite=out["grad_is_finite"],
                grads=[b.grad.detach().cpu().clone() for b in tr.banks], params=[b.flat.detach().cpu().clone() for b in tr.banks],
                probs=out["probs"].clone())


def main():
    experiment, out_dir = sys.argv[1], sys.argv[2]
    rank, world = int(os.environ["RANK"]), int(os.environ["WORLD_SIZE"])
    torch.cuda.set_device(0)
    dev = torch.device("cuda", 0)
    dist.init_process_group("gloo", rank=rank, world_size=world)
    tr = build(experiment, dev, rank, world)
    assert tr.collectives
    noises = global_noises(world)[rank * B_PER_RANK:(rank + 1) * B_PER_RANK]
    out = tr.train_step(U.tiny_tokens(), noises, S)
    torch.cuda.synchronize()
    torch.save(snapshot(tr, out), os.path.join(out_dir, f"rank{rank}.pt"))
    dist.barrier()
    dist.destroy_process_group()


if __name__ == "__main__":
    try:
        main()
    except BaseException:
        import traceback
        with open(os.path.join(sys.argv[2], f"rank{os.environ.get('RANK', '0')}.err"), "w") as f:     # the launcher's own traceback hides the child's
            traceback.print_exc(file=f)
        raise
