"""exp-2: the trainable prompt-prefix tokens (``FairEmbeddings`` + ``expand_tokenizer``, exp-2-debias-gender-token/1-main-debias.py:86-146,
:919-935).  The reference appends ``train_num_tokens`` placeholder tokens to the vocabulary, copies the embedding rows of as many randomly
chosen existing tokens into them, and optimises ONE fp32 table ``token_embedding.weight`` [n+1, D] (row 0 = zeros, the "not a prefix token"
slot of ``fair_token_id_map``; rows 1..n = the prefix vectors).  The prompt ``"".join(prefix_tokens) + prompt`` carries the placeholders
right after BOS; ``FairEmbeddings.forward`` swaps ``table[k] + position_embedding`` in at those positions, everything else of the text
encoder and the U-Net stays frozen (:946).

Here the table lives in a ``ParamBank`` (one flat fp32 buffer + grad / Adam / EMA twins) so that the trainer's gradient all-reduce,
finite guard, AdamW and EMA launches treat it exactly like a LoRA bank.
"""
import torch

from .layers import F32, ParamBank

KEY = "token_embedding.weight"


class PrefixEmbedding:
    def __init__(self, text_encoder, n, device, seed=0, state_dict=None):
        """``text_encoder``: the (frozen) CLIPTextModel whose vocabulary rows initialise the prefix (``expand_tokenizer`` :124-146: a
        shuffled list of existing token ids, first n).  ``state_dict``: a ``FairEmbeddings`` state dict (``prefix_embedding.pth``)."""
        self.n = int(n)
        self.device = device
        D = text_encoder.tok.shape[1]
        self.bank = ParamBank({KEY: (self.n + 1, D)}, device)
        self._pos = text_encoder.pos
        if state_dict is not None:
            w = state_dict[KEY] if isinstance(state_dict, dict) else state_dict
            if tuple(w.shape) != (self.n + 1, D):
                raise ValueError(f"prefix embedding: {tuple(w.shape)} != {(self.n + 1, D)} (train_num_tokens + 1, hidden size)")
            self.bank.view(KEY).copy_(w.to(device, F32))
        else:
            g = torch.Generator().manual_seed(seed)
            rows = torch.randperm(text_encoder.tok.shape[0], generator=g)[:self.n].to(device)
            self.bank.view(KEY)[1:].copy_(text_encoder.tok[rows].to(F32))       # row 0 stays zero (:95-96)
        self.bank.ema.copy_(self.bank.flat)

    @property
    def weight(self):
        return self.bank.view(KEY)

    def vectors(self, ema=False):
        """[n, D] fp32: what replaces the token embedding at positions 1..n of the debiased prompt (the position embedding is added by the
        text encoder, as ``FairEmbeddings.forward`` does, :116-120)."""
        return self.bank.view(KEY, self.bank.ema if ema else None)[1:]

    def state_dict(self, ema=False):
        """Same three entries as ``FairEmbeddings.state_dict()`` (``2-export-checkpoint.py:566-575``)."""
        P = self._pos.shape[0]
        return {"position_ids": torch.arange(P).expand((1, -1)).clone(), "position_embedding.weight": self._pos.detach().float().cpu().clone(),
                KEY: self.bank.view(KEY, self.bank.ema if ema else None).detach().cpu().clone()}

    def load_state_dict(self, sd, strict=False):
        self.bank.load_state_dict({KEY: sd[KEY]}, strict=True)
