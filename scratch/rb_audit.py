"""Audit of gemm_rb.hip's generated code (hipcc -save-temps .s): the B-ring loads are hidden from the compiler, so any compiler-generated instruction
(outside ;;#ASMSTART / ;;#ASMEND) that READS a register some inline-asm global_load writes -- a spill, a v_mov, a v_accvgpr_* copy -- other than an MFMA is a
potential use of a register whose load is still in flight.    python scratch/rb_audit.py <file.s> [kernel substring]"""
import re, sys
src = open(sys.argv[1]).read().split("\n")
want = sys.argv[2] if len(sys.argv) > 2 else "gemm_rb_kernel"
def regs(tok):
    out = set()
    for m in re.finditer(r"\b([va])\[(\d+):(\d+)\]", tok):
        out |= {(m.group(1), i) for i in range(int(m.group(2)), int(m.group(3)) + 1)}
    for m in re.finditer(r"\b([va])(\d+)\b", tok):
        out.add((m.group(1), int(m.group(2))))
    return out
kern = None
inasm = False
ring = {}
body = {}
for ln, l in enumerate(src):
    m = re.match(r"^(_Z\w+):", l)
    if m:
        kern = m.group(1) if want in m.group(1) else None
        continue
    if kern is None:
        continue
    if "s_endpgm" in l:
        kern = None
        continue
    if "#ASMSTART" in l:
        inasm = True
        continue
    if "#ASMEND" in l:
        inasm = False
        continue
    t = l.strip()
    if not t or t.startswith(";") or t.startswith("."):
        continue
    body.setdefault(kern, []).append((ln + 1, inasm, t))
    if inasm and t.startswith("global_load_dwordx4"):
        ring.setdefault(kern, set()).update(regs(t.split(",")[0]))
for k, ins in body.items():
    R = ring.get(k, set())
    bad = []
    for ln, ia, t in ins:
        if ia:
            continue
        op = t.split()[0]
        ops = t[len(op):].split(",")
        if op.startswith("v_mfma"):
            continue
        # destination is the first operand for everything but stores
        srcs = ops if op.startswith("scratch_store") or op.startswith("global_store") or op.startswith("ds_write") else ops[1:]
        dst = [] if op.startswith("scratch_store") or op.startswith("global_store") or op.startswith("ds_write") else ops[:1]
        rs = set().union(*[regs(o) for o in srcs]) if srcs else set()
        ws = set().union(*[regs(o) for o in dst]) if dst else set()
        if rs & R:
            bad.append((ln, "READS ring", t))
        elif ws & R:
            bad.append((ln, "writes ring", t))
    nsp = sum(1 for _, ia, t in ins if not ia and t.startswith("scratch_"))
    print(k, "ring registers:", len(R), "compiler scratch ops:", nsp, "compiler instructions touching ring registers:", len(bad))
    for b in bad[:25]:
        print("   ", b)
