"""CPU oracle for the fairness-finetuning hot path (TEST INFRASTRUCTURE ONLY).

This package is a plain-PyTorch fp32 restatement of the arithmetic on the
reference's distributional-alignment training step.  It is
the *checker*: only ``tests/``, ``__graft_entry__.smoke()`` and the
``cpu_baseline`` leg of ``bench.py`` may import it.  The product package
(``finetune_fair_diffusion_amd``) never imports, links or falls back to it.

Pinning status:

* reference-owned pure logic (``expand_bbox``, ``generate_dynamic_targets``,
  ``gen_dynamic_weights``, ``apply_grad_hook_face``, ``get_face_gender`` scatter,
  ``parse_args`` + YAML overlay): PINNED -- golden vectors in ``tests/golden`` were
  produced by executing the reference's own source lifted from
  ``exp-1-debias-gender/1-main-debias.py`` (``tests/golden/make_golden.py``).
* third-party arithmetic (diffusers==0.19.3 UNet2DConditionModel /
  LoRAAttnProcessor / DPMSolverMultistepScheduler / AutoencoderKL / EMAModel,
  transformers==4.30.0 CLIPTextModel, torchvision==0.16.2 mobilenet_v3_large):
  those packages are pinned in ``environment.yml:118-203`` but are neither vendored
  under /root/reference nor installed, and the reference holds no tests or golden
  vectors for them.  Their published algorithms are restated here: PARITY UNPINNED
  for these pieces; they are self-validated by analytic invariants (tests/).
"""
