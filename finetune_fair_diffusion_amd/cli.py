"""Command-line / YAML surface of the reference's ``1-main-debias.py`` (``parse_args``,
exp-1-debias-gender/1-main-debias.py:327-644): the same 50 flags with the same defaults and the same
YAML-overlay rule -- ``args[key] = type(args[key])(value)`` (:632-638), so keys whose default is None
cannot be set from YAML and booleans follow Python's ``bool(value)`` -- plus the build's own additions
(prefixed below), which the reference does not have:

  --num_denoising_steps  fixed S instead of ``random.choices(range(19,24))`` (:1779)
  --synthetic            synthetic weights / token ids / face provider (no network, no data.zip)
  --face_provider        detector seam: "synthetic" (default) or "detector" (insightface + face_recognition, when installed)
  --num_classifier_logits  80 (exp-1) / 6 (exp-3,5) / 8 (exp-4)
  --lora_up_std          std of the LoRA ``up`` init; 0 (default) = zeros like the reference, non-zero only for synthetic experiments

The multi-attribute experiments change a few flags and defaults (exp-3-debias-gender-race/1-main-debias.py:343-660,
exp-4-debias-gender-race-age/...:343-672, exp-5-...:343-690; exp-2-debias-gender-token/...:453-780 drops the two LoRA switches and adds
``--train_num_tokens``): ``factor{1,2}`` split per attribute,
``face_gender[_race[_age]]_confidence_level``, bigger batches, three more prompt files in exp-5 -- ``EXPERIMENT_CLI``.

Pinned by tests/golden/reference_cli.json and reference_cli_multi.json (defaults and every YAML overlay of
exp-1/2/3/4/5, produced by running the reference's own parse_args).
"""
import argparse
import os

import yaml


_FF = "../data/2-trained-classifiers/fairface_MobileNetLarge_"
_MULTI = dict(max_train_steps=15000, weight_loss_face=0.1, uncertainty_threshold=0.4, val_images_per_prompt_GPU=24,
              face_feats_path="../data/3-face-features/FairFace_MobileNetLarge_GenderRace4_09041634/face_feats.pkl")
_GR = dict(factor1_gender=0.2, factor1_race=0.6, factor2_gender=0.2, factor2_race=0.3)
# experiment -> (changed defaults, removed flags, added float flags, added str flags)
EXPERIMENT_CLI = {
    "exp-1": ({}, [], {}, {}),
    # exp-2 (prefix-token tuning, exp-2-debias-gender-token/1-main-debias.py:453-780): no LoRA switches, one int flag (below)
    "exp-2": ({}, ["train_text_encoder", "train_unet"], {}, {}),
    "exp-3": (dict(_MULTI, train_images_per_prompt_GPU=16, classifier_weight_path=_FF + "GenderRace4_09041216/epoch=9-step=3380_MobileNetLarge.pt"),
              ["factor1", "factor2", "face_gender_confidence_level"], dict(_GR, face_gender_race_confidence_level=0.8), {}),
    "exp-4": (dict(_MULTI, train_images_per_prompt_GPU=20, classifier_weight_path=_FF + "GenderRace4Age2_09151907/epoch=9-step=3380_MobileNetLarge.pt"),
              ["factor1", "factor2", "face_gender_confidence_level"],
              dict(_GR, factor1_age=0.6, factor2_age=0.3, face_gender_race_age_confidence_level=0.75), {}),
    "exp-5": (dict(_MULTI, train_images_per_prompt_GPU=16, classifier_weight_path=_FF + "GenderRace4_09041216/epoch=9-step=3380_MobileNetLarge.pt"),
              ["factor1", "factor2", "face_gender_confidence_level"], dict(_GR, face_gender_race_confidence_level=0.8),
              dict(prompt_occupation_w_style_and_context_path="../data/1-prompts/occupation_w_style_and_context.json",
                   prompt_personal_descroptor_path="../data/1-prompts/personal_descriptor.json",   # (sic) the reference's spelling
                   prompt_sports_path="../data/1-prompts/sports.json")),
}


def build_parser(experiment="exp-1"):
    p = argparse.ArgumentParser(description="Script to finetune Stable Diffusion for debiasing purposes.")
    changed, removed, add_f, add_s = EXPERIMENT_CLI[experiment]

    def a(flag, **kw):
        name = flag.lstrip("-")
        if name in removed:
            return
        if name in changed:
            kw["default"] = changed[name]
        p.add_argument(flag, **kw)
    for k, v in add_f.items():
        p.add_argument("--" + k, type=float, default=v)
    for k, v in add_s.items():
        p.add_argument("--" + k, type=str, default=v)
    if experiment == "exp-2":
        p.add_argument("--train_num_tokens", type=int, default=5)       # number of tokens to finetune as prompt prefix (:479-484)
    # 1. experiment setting
    a("--proj_name", default="debias-SD", type=str)
    a("--pretrained_model_name_or_path", type=str, default="runwayml/stable-diffusion-v1-5")
    a("--train_text_encoder", action="store_true", default=True)
    a("--train_unet", action="store_true", default=False)
    a("--seed", type=int, default=5991)
    a("--max_train_steps", type=int, default=10000)
    a("--checkpointing_steps", type=int, default=20)
    a("--checkpoints_total_limit", type=int, default=2)
    a("--checkpointing_steps_long", type=int, default=200)
    a("--resume_from_checkpoint", type=str, default=None)
    a("--mixed_precision", type=str, default="fp16", choices=["no", "fp16", "bf16"])
    a("--rank", type=int, default=50)
    a("--train_plot_every_n_iter", type=int, default=20)
    a("--evaluate_every_n_iter", type=int, default=200)
    a("--guidance_scale", type=float, default=7.5)
    a("--EMA_decay", type=float, default=0.996)
    # 2. loss weights
    a("--weight_loss_img", type=float, default=8)
    a("--weight_loss_face", type=float, default=1)
    a("--uncertainty_threshold", type=float, default=0.2)
    a("--factor1", type=float, default=0.2)
    a("--factor2", type=float, default=0.2)
    # 3. batch sizes
    a("--train_images_per_prompt_GPU", type=int, default=8)
    a("--train_GPU_batch_size", type=int, default=4)
    a("--val_images_per_prompt_GPU", type=int, default=8)
    a("--val_GPU_batch_size", type=int, default=8)
    # 4. data / external files
    a("--prompt_occupation_path", type=str, default="../data/1-prompts/occupation.json")
    a("--classifier_weight_path", type=str, default="../data/2-trained-classifiers/CelebA_MobileNetLarge_08060852/epoch=9-step=12660_MobileNetLarge.pt")
    a("--face_feats_path", type=str, default="../data/3-face-features/CelebA_MobileNetLarge_08240859/face_feats.pkl")
    a("--opensphere_config", type=str, default="../data/4-opensphere_checkpoints/opensphere_checkpoints/20220424_210641/config.yml")
    a("--opensphere_model_path", type=str, default="../data/4-opensphere_checkpoints/opensphere_checkpoints/20220424_210641/models/backbone_100000.pth")
    a("--output_dir", type=str, default="./outputs")
    a("--logging_dir", type=str, default="logs")
    a("--report_to", type=str, default="wandb")
    # 5. optimisation
    a("--learning_rate", type=float, default=5e-5)
    a("--lr_scheduler", type=str, default="constant")
    a("--lr_warmup_steps", type=int, default=0)
    a("--lr_num_cycles", type=int, default=1)
    a("--lr_power", type=float, default=1.0)
    a("--allow_tf32", action="store_true", default=True)
    a("--adam_beta1", type=float, default=0.9)
    a("--adam_beta2", type=float, default=0.999)
    a("--adam_weight_decay", type=float, default=1e-2)
    a("--adam_epsilon", type=float, default=1e-08)
    a("--max_grad_norm", default=100.0, type=float)
    a("--img_size_small", type=int, default=224)
    a("--size_face", type=int, default=224)
    a("--size_aligned_face", type=int, default=112)
    a("--face_gender_confidence_level", type=float, default=0.9)
    a("--local_rank", type=int, default=-1)
    a("--config", type=str, default=None)
    return p


EXTRA_DEFAULTS = dict(num_denoising_steps=0, synthetic=False, face_provider="synthetic", num_classifier_logits=80, lora_up_std=0.0)


def parse_args(input_args=None, with_extras=False, experiment="exp-1"):
    p = build_parser(experiment)
    if with_extras:
        p.add_argument("--num_denoising_steps", type=int, default=0, help="0 = draw from range(19,24) like the reference")
        p.add_argument("--synthetic", action="store_true", default=False)
        p.add_argument("--face_provider", type=str, default="synthetic")
        p.add_argument("--num_classifier_logits", type=int, default=80)
        p.add_argument("--lora_up_std", type=float, default=0.0)
    args = p.parse_args(input_args) if input_args is not None else p.parse_args()
    if args.config:
        with open(args.config, "r") as f:
            config_data = yaml.safe_load(f)
        d = vars(args)
        for key, value in config_data.items():
            d[key] = type(d[key])(value)
        args = argparse.Namespace(**d)
    env_local_rank = int(os.environ.get("LOCAL_RANK", -1))
    if env_local_rank != -1 and env_local_rank != args.local_rank:
        args.local_rank = env_local_rank
    return args
