"""Tests of the e4m3 self-attention forward (scratch/attn_fp8_experiment.hip) as they stood when the kernel left the product in round 5
(no in-step win -- profiles/r03_bench_bf16_fp8attn_v2_not_headline.json 5.89 images/s against 5.9 for plain bf16).  Not collected by pytest."""

@pytest.mark.parametrize("B,H,T,d", [(2, 8, 1024, 40), (1, 8, 4096, 40), (2, 8, 256, 80), (2, 4, 256, 160), (1, 2, 64, 160)])
def test_attention_fwd_fp8_band(ops, dev, B, H, T, d):
    """BASELINE configs[4]: e4m3 QK^T / PV self-attention forward (per-row Q scale, per-64-key-tile K / V scales, fp32 softmax).
    The reference never ran fp8, so acceptance is a stated BAND on the hardest input (white-noise q, k, v: no structure for the 3-bit
    significand to exploit; measured 6.6e-2 .. 1.0e-1 max, 5e-2 RMS, LSE 3.6e-2): O within 1.5e-1 of max|O| and 7e-2 relative RMS of the
    fp32 reference (the 16-bit kernel sits at 3e-3), LSE within 5e-2 absolute; the same band vs the 16-bit path of this library.
    In the network the effect is an order of magnitude smaller: SD-v1.5 U-Net eps with e4m3 self-attention at all four levels differs
    from the fp32 oracle by 1.3e-2 max / 9.5e-3 RMS (tests/run_bf16_checks.py::sd15_unet)."""
    C = H * d
    q, k, v = rnd(B, T, C, dev=dev, seed=1), rnd(B, T, C, dev=dev, seed=2), rnd(B, T, C, dev=dev, seed=3)
    oref, lref = _attn_ref(q.float(), k.float(), v.float(), H)
    o8, lse8 = ops.attn_fwd_fp8(q.reshape(B * T, C), k.reshape(B * T, C), v.reshape(B * T, C), B, H, T, d, need_lse=True)
    vt = ops.transpose_btc(v.reshape(B * T, C), B, T, C)
    o16, lse16 = ops.attn_fwd(q.reshape(B * T, C), k.reshape(B * T, C), vt, B, H, T, T, d, 1, need_lse=True)
    rms = float((o8.reshape(B, T, C).float() - oref).pow(2).mean().sqrt() / oref.pow(2).mean().sqrt())
    print(f"fp8 attention d={d} T={T}: rel RMS err vs fp32 {rms:.3e}; max|lse err| {float((lse8 - lref).abs().max()):.3e}")
    check("attn fp8 fwd vs fp32", o8.reshape(B, T, C), oref, 1.5e-1)
    check("attn fp8 fwd vs 16-bit path", o8.reshape(B, T, C), o16.reshape(B, T, C).float(), 1.5e-1)
    assert rms < 7e-2              # RMS error relative to the RMS of O
    assert float((lse8 - lref).abs().max()) < 5e-2
