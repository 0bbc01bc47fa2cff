"""MI355X-native mirror of transformers ``CLIPTextModel`` with the diffusers text-encoder LoRA patch
(``PatchedLoraProjection`` on q/k/v/out_proj + fc1/fc2; reference :829-883) for the calls
``text_encoder(input_ids, attention_mask)[0]`` (exp-1-debias-gender/1-main-debias.py:1011-1014,
:1029-1032, :1078-1081, :1096-1099).

The reference feeds N identical prompts (``[prompt] * N`` :1005/:1072) and N identical empty prompts,
so the hidden states of a rollout are 2 distinct sequences; ``encode_pair`` runs exactly those two.
Forward and (for text-encoder LoRA training) an explicit backward from d(prompt_embeds).
"""
import torch

from . import ops
from .layers import F16, F32, Linear, LoRAPair, Norm, ParamBank, lora_linear_bwd, lora_linear_fwd
from .weights import TE_LORA_TARGETS, CLIPTextConfig, clip_lora_param_shapes, clip_param_shapes


class CLIPTextModel:
    def __init__(self, cfg: CLIPTextConfig, state_dict, device):
        self.config, self.device = cfg, device
        sd, dev = state_dict, device
        missing = [k for k in clip_param_shapes(cfg) if k not in sd]
        if missing:
            raise KeyError(f"text encoder state_dict is missing {len(missing)} tensors, e.g. {missing[:3]}")
        self.tok = sd["text_model.embeddings.token_embedding.weight"].to(dev, F16)
        self.pos = sd["text_model.embeddings.position_embedding.weight"].to(dev, F16)
        self.layers = []
        for i in range(cfg.num_hidden_layers):
            p = f"text_model.encoder.layers.{i}."
            self.layers.append(dict(
                ln1=Norm(sd, p + "layer_norm1", dev), ln2=Norm(sd, p + "layer_norm2", dev),
                q=Linear(sd, p + "self_attn.q_proj", dev), k=Linear(sd, p + "self_attn.k_proj", dev),
                v=Linear(sd, p + "self_attn.v_proj", dev), o=Linear(sd, p + "self_attn.out_proj", dev),
                fc1=Linear(sd, p + "mlp.fc1", dev), fc2=Linear(sd, p + "mlp.fc2", dev), lora={}))
        self.final_ln = Norm(sd, "text_model.final_layer_norm", dev)
        self.lora_bank = None
        self._ctx = None

    # ------------------------------------------------------------------ LoRA
    def add_lora(self, rank, state_dict=None, seed=1):
        shapes = clip_lora_param_shapes(self.config, rank)
        self.lora_bank = ParamBank(shapes, self.device)
        if state_dict is None:
            from .weights import synthetic_state_dict
            state_dict = synthetic_state_dict(shapes, seed=seed)
        self.lora_bank.load_state_dict(state_dict)
        short = dict(zip(TE_LORA_TARGETS, ("q", "k", "v", "o", "fc1", "fc2")))
        for i, L in enumerate(self.layers):
            for tgt, key in short.items():
                p = f"text_model.encoder.layers.{i}.{tgt}.lora_linear_layer."
                L["lora"][key] = LoRAPair(self.lora_bank, p + "down.weight", p + "up.weight")
        self.refresh_lora()
        return self.lora_bank

    def refresh_lora(self):
        from .layers import refresh_pairs
        refresh_pairs([lo for L in self.layers for lo in L["lora"].values()])

    def load_state_dict(self, sd, strict=False):
        if self.lora_bank is not None:
            self.lora_bank.load_state_dict(sd, strict=strict)
            self.refresh_lora()

    # ------------------------------------------------------------------ forward / backward
    def forward(self, input_ids, attention_mask=None, record=False, prefix=None):
        """input_ids [B,T] int64, attention_mask [B,T] (1 = keep).  Returns (last_hidden_state [B,T,D] fp16,).
        ``prefix`` = (row, vectors [n, D]): the token embeddings of positions 1..n of sequence ``row`` are replaced by ``vectors``
        (exp-2's learned prefix tokens, ``FairEmbeddings.forward`` gen-images.py:72-90: fair token embedding + position embedding)."""
        cfg = self.config
        B, T = input_ids.shape
        D, H = cfg.hidden_size, cfg.num_attention_heads
        d = D // H
        vocab = self.tok.shape[0]
        oov = input_ids >= vocab            # host tensor (token ids arrive from the tokenizer on the CPU): no device sync
        if prefix is not None:
            # exp-2: the placeholder ids of the n prefix tokens lie beyond the vocabulary; they must sit exactly at positions 1..n of
            # ``row`` (their embedding rows are overwritten below) -- any other out-of-vocabulary id is a corrupt input, not a placeholder
            row, vec = prefix
            allowed = torch.zeros_like(oov)
            allowed[row, 1:1 + vec.shape[0]] = True
            if bool((oov & ~allowed).any()) or int(input_ids.min()) < 0:
                raise ValueError("CLIPTextModel.forward: token id outside the vocabulary at a position that is not a prefix placeholder")
            ids = input_ids.clamp(max=vocab - 1).to(self.device)
            te = self.tok[ids].clone()
            te[row, 1:1 + vec.shape[0]] = vec.to(self.device, te.dtype)
        else:
            if bool(oov.any()) or int(input_ids.min()) < 0:
                raise ValueError(f"CLIPTextModel.forward: token id outside the vocabulary of {vocab} entries")
            te = self.tok[input_ids.to(self.device)]
        x = (te + self.pos[:T][None]).reshape(B * T, D).contiguous()  # embedding gather: plumbing
        kv = attention_mask.to(self.device, torch.int32).contiguous() if attention_mask is not None else None
        ctx = [] if record else None
        for L in self.layers:
            lo = L["lora"]
            n1, s1 = ops.layernorm(x, L["ln1"].gamma, L["ln1"].beta, cfg.layer_norm_eps, save_stats=True)
            q, tq = lora_linear_fwd(n1, L["q"], lo.get("q"))
            k, tk = lora_linear_fwd(n1, L["k"], lo.get("k"))
            v, tv = lora_linear_fwd(n1, L["v"], lo.get("v"))
            if record:
                a, P = ops.small_attn_fwd(q, k, v, kv, B, H, T, d, d ** -0.5, True, save_p=True)
            else:
                a, P = ops.small_attn_fwd(q, k, v, kv, B, H, T, d, d ** -0.5, True), None
            h1, to = lora_linear_fwd(a, L["o"], lo.get("o"), residual=x)
            n2, s2 = ops.layernorm(h1, L["ln2"].gamma, L["ln2"].beta, cfg.layer_norm_eps, save_stats=True)
            if record:
                z, t1 = lora_linear_fwd(n2, L["fc1"], lo.get("fc1"))
                m = ops.act_fwd(z, "quick_gelu")
            else:
                m, t1 = lora_linear_fwd(n2, L["fc1"], lo.get("fc1"), act="quick_gelu")
                z = None
            h2, t2 = lora_linear_fwd(m, L["fc2"], lo.get("fc2"), residual=h1)
            if record:
                ctx.append(dict(x=x, s1=s1, n1=n1, tq=tq, tk=tk, tv=tv, q=q, k=k, v=v, P=P, a=a, to=to, h1=h1, s2=s2, n2=n2, z=z, t1=t1, m=m, t2=t2))
            x = h2
        y, sf = ops.layernorm(x, self.final_ln.gamma, self.final_ln.beta, cfg.layer_norm_eps, save_stats=True)
        if record:
            self._ctx = dict(layers=ctx, x=x, sf=sf, B=B, T=T)
        return (y.view(B, T, D),)

    __call__ = forward

    def backward(self, d_out, gscale):
        """d_out: [B,T,D] fp16 = gscale * dL/d(last_hidden_state).  Accumulates the LoRA gradients; returns gscale * dL/d(input embeddings)
        [B,T,D] fp16 (token + position embedding sum: what exp-2's prefix vectors enter through)."""
        cfg, c = self.config, self._ctx
        B, T = c["B"], c["T"]
        D, H = cfg.hidden_size, cfg.num_attention_heads
        d = D // H
        dx = ops.layernorm_bwd(c["x"], d_out.reshape(B * T, D).contiguous(), self.final_ln.gamma, c["sf"])
        for L, s in zip(reversed(self.layers), reversed(c["layers"])):
            lo = L["lora"]
            dm = lora_linear_bwd(dx, s["m"], s["t2"], L["fc2"], lo.get("fc2"), gscale)
            dz = ops.act_bwd(s["z"], dm, "quick_gelu")
            dn2 = lora_linear_bwd(dz, s["n2"], s["t1"], L["fc1"], lo.get("fc1"), gscale)
            dh1 = ops.layernorm_bwd(s["h1"], dn2, L["ln2"].gamma, s["s2"], add=dx)
            da = lora_linear_bwd(dh1, s["a"], s["to"], L["o"], lo.get("o"), gscale)
            dq, dk, dv = ops.small_attn_bwd(s["q"], s["k"], s["v"], s["P"], da, B, H, T, d, d ** -0.5)
            dn1 = lora_linear_bwd(dq, s["n1"], s["tq"], L["q"], lo.get("q"), gscale)
            dn1 = lora_linear_bwd(dk, s["n1"], s["tk"], L["k"], lo.get("k"), gscale, residual=dn1)
            dn1 = lora_linear_bwd(dv, s["n1"], s["tv"], L["v"], lo.get("v"), gscale, residual=dn1)
            dx = ops.layernorm_bwd(s["x"], dn1, L["ln1"].gamma, s["s1"], add=dh1)
        self._ctx = None
        return dx.view(B, T, D)
