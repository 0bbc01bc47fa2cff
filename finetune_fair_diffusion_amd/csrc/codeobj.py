"""Static inspection of the gfx950 code objects inside a built libfairdiff_hip*.so (build-time / test-time tool, no GPU needed):
splits the library's .hip_fatbin into its per-translation-unit offload bundles, unbundles the gfx950 ELF of each and returns the
disassembly and the kernel resource notes.  Used by tests/test_cpu.py to hold two build invariants: no packed-fp32 VALU instruction in any
shipped kernel ("the round-3 hazard"), and the register / scratch budgets of the hot kernels."""
import os
import re
import subprocess
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"
MAGIC = b"__CLANG_OFFLOAD_BUNDLE__"


def code_objects(lib_path):
    """-> list of (index, path of the gfx950 ELF) extracted into a temporary directory (kept for the life of the process)."""
    tmp = tempfile.mkdtemp(prefix="fd_codeobj_")
    fat = os.path.join(tmp, "fat.bin")
    subprocess.run([f"{LLVM}/llvm-objcopy", "--dump-section", f".hip_fatbin={fat}", lib_path, os.path.join(tmp, "discard.so")], check=True)
    data = open(fat, "rb").read()
    idx, i = [], data.find(MAGIC)
    while i >= 0:
        idx.append(i)
        i = data.find(MAGIC, i + 1)
    idx.append(len(data))
    out = []
    for k in range(len(idx) - 1):
        b = os.path.join(tmp, f"b{k}.bin")
        open(b, "wb").write(data[idx[k]:idx[k + 1]])
        targets = subprocess.run([f"{LLVM}/clang-offload-bundler", "--list", "--type=o", f"--input={b}"], capture_output=True, text=True, check=True).stdout.split()
        t = [x for x in targets if "gfx950" in x]
        if not t:
            continue
        co = os.path.join(tmp, f"k{k}.co")
        subprocess.run([f"{LLVM}/clang-offload-bundler", "--unbundle", "--type=o", f"--input={b}", f"--targets={t[0]}", f"--output={co}"], check=True)
        out.append((k, co))
    return out


def disassembly(co):
    return subprocess.run([f"{LLVM}/llvm-objdump", "-d", co], capture_output=True, text=True, check=True).stdout


def kernel_resources(co):
    """-> {kernel name: dict(vgpr=, agpr=, sgpr=, scratch=, lds=)} from the code object's metadata notes."""
    notes = subprocess.run([f"{LLVM}/llvm-readelf", "--notes", co], capture_output=True, text=True, check=True).stdout
    res = {}
    for blk in notes.split("- .agpr_count:")[1:]:
        def f(key, blk=blk):
            m = re.search(rf"\.{key}:\s+(\S+)", blk)
            return m.group(1) if m else None
        res[f("name")] = dict(agpr=int(blk.split()[0]), vgpr=int(f("vgpr_count")), sgpr=int(f("sgpr_count")), scratch=int(f("private_segment_fixed_size")),
                              lds=int(f("group_segment_fixed_size")))
    return res


PACKED_F32 = re.compile(r"\bv_pk_(add|mul|fma)_f32\b")


def packed_f32_sites(lib_path):
    """-> list of (kernel symbol, instruction text) for every packed-fp32 VALU instruction in the library's gfx950 code."""
    sites = []
    for _, co in code_objects(lib_path):
        cur = None
        for ln in disassembly(co).splitlines():
            m = re.match(r"^[0-9a-f]+ <(\S+)>:", ln)
            if m:
                cur = m.group(1)
            elif PACKED_F32.search(ln):
                sites.append((cur, ln.strip().split("//")[0].strip()))
    return sites


if __name__ == "__main__":
    import sys
    for lib in sys.argv[1:]:
        s = packed_f32_sites(lib)
        print(f"{lib}: {len(s)} packed-fp32 VALU instructions" + (f", e.g. {s[0]}" if s else ""))
        for _, co in code_objects(lib):
            for n, r in kernel_resources(co).items():
                if r["scratch"]:
                    print(f"   scratch {r['scratch']:4d} B  vgpr {r['vgpr']:3d}  {n}")
