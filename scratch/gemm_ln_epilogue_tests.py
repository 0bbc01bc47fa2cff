"""Tests of the LayerNorm-as-second-output epilogue (scratch/gemm_ln_epilogue_experiment.h) as they stood when it left the product in round 5.  Not collected by pytest."""

@pytest.mark.parametrize("M,N,K,fused", [(32768, 320, 320, True), (65536, 320, 320, True), (12800, 320, 1280, True), (10250, 320, 320, True),
                                           (4096, 320, 320, False), (16384, 640, 640, False)])
def test_gemm_layernorm_epilogue(ops, dev, M, N, K, fused, monkeypatch):
    """fd_gemm_desc.ln_out: the GEMM whose tile holds whole rows writes LayerNorm(row) as a second output.  C itself is
    bit-identical to the launch without it; the normalised copy and the saved statistics against torch and against fd_layernorm_fwd on the same
    C; M tails; shapes whose kernel cannot (small M, N != 320) fall back to the standalone pass inside ops.gemm."""
    import ctypes
    from finetune_fair_diffusion_amd import lib
    monkeypatch.setattr(ops, "LN_EPILOGUE", True)          # off by default in the product (profiles/r04_layernorm_epilogue.txt); FD_LN_EPILOGUE=1 turns it on
    a, b = rnd(M, K, dev=dev, seed=1), rnd(N, K, dev=dev, scale=0.1, seed=2)
    a2, b2 = rnd(M, 8, dev=dev, seed=3), rnd(N, 8, dev=dev, seed=4)
    bias, res = rnd(N, dev=dev, dtype=torch.float32, seed=5), rnd(M, N, dev=dev, seed=6) * 3 + 0.5
    gamma = rnd(N, dev=dev, dtype=torch.float32, seed=7) * 0.2 + 1
    beta = rnd(N, dev=dev, dtype=torch.float32, seed=8) * 0.2
    d = lib.GemmDesc(); d.M, d.N, d.K, d.K2, d.batch, d.ldc, d.alpha, d.ldr = M, N, K, 8, 1, N, 1.0, N
    d.A2 = d.residual = 1 << 20
    ws = ops.gemm_workspace()
    d.workspace, d.workspace_bytes = ws.data_ptr(), ws.numel() * 4
    assert bool(lib.get().fd_gemm_ln_ok(ctypes.byref(d))) == fused
    plain = ops.gemm(a, b, a2=a2, b2=b2, bias=bias, residual=res)
    c, n, st = ops.gemm(a, b, a2=a2, b2=b2, bias=bias, residual=res, ln=(gamma, beta, 1e-5))
    assert torch.equal(plain, c)
    ref = F.layer_norm(c.float(), (N,), gamma, beta, 1e-5)
    check("LayerNorm from the GEMM epilogue", n, ref, 2e-3)
    mean = c.double().mean(1)
    rstd = 1.0 / torch.sqrt(c.double().var(1, unbiased=False) + 1e-5)
    check("saved mean", st[:, 0], mean, 1e-5)
    check("saved rstd", st[:, 1], rstd, 1e-5)
    n0, st0 = ops.layernorm(c, gamma, beta, 1e-5, save_stats=True)
    assert float((n.float() - n0.float()).abs().max()) <= 2e-3 * float(n0.float().abs().max())
    c2, n2, st2 = ops.gemm(a, b, a2=a2, b2=b2, bias=bias, residual=res, ln=(gamma, beta, 1e-5))
    assert torch.equal(n, n2) and torch.equal(st, st2)
    if fused and M >= 32768:                      # rows do not depend on the tile: the first rows alone (128-row tiles) give the same bits
        Ms = 12800
        cs, ns, sts = ops.gemm(a[:Ms], b, a2=a2[:Ms], b2=b2, bias=bias, residual=res[:Ms], ln=(gamma, beta, 1e-5))
        assert torch.equal(ns, n[:Ms]) and torch.equal(sts, st[:Ms])



def test_sd15_unet_with_the_layernorm_epilogue_equals_the_separate_pass(full, dev, monkeypatch):
    """fd_gemm_desc.ln_out inside the real network (off by default; FD_LN_EPILOGUE=1): at batch 4 the 64^2-level proj_in / attn1.to_out / attn2.to_out
    GEMMs (M = 16384, N = 320) are eligible and write norm1 / norm2 / norm3 themselves.  Forward and LoRA gradient against the separate-pass run of the
    same network: equal to the fp16 rounding of the normalised activations (the statistics differ in summation order and in the variance formula)."""
    from finetune_fair_diffusion_amd import ops
    om, pm = full
    enc = _pair_embeddings(om, dev)
    unet_p = pm["unet"]
    unet_p.prepare_timesteps([601])
    x = torch.randn(4, 4, 64, 64, generator=torch.Generator().manual_seed(7)).to(dev)
    g = torch.randn(8, 4, 64, 64, generator=torch.Generator().manual_seed(8)).to(dev)
    outs = []
    for on in (False, True):
        monkeypatch.setattr(ops, "LN_EPILOGUE", on)
        unet_p.prepare_prompt(enc.to(dev).half(), record=True)
        eps = unet_p.forward_step(x, 0, record=True, pair=True)
        bank = unet_p.lora_bank
        bank.grad.zero_()
        unet_p.backward_step(g * 64.0, 64.0)
        unet_p.finish_prompt_backward(64.0, need_denc=False)
        outs.append((eps.clone(), bank.grad.clone()))
    check("eps: LayerNorm epilogue vs separate pass", outs[1][0], outs[0][0], 4e-3)
    cos = float(F.cosine_similarity(outs[1][1].double().flatten(), outs[0][1].double().flatten(), dim=0))
    print("cosine(LoRA grads, LayerNorm epilogue vs separate pass) =", cos)
    assert cos > 0.9999


