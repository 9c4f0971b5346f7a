"""Static instruction mix of one kernel of a built library (every code object inside it is searched: one per translation unit).

    python tools/isa_stats.py <lib.so> <kernel-symbol-substring> [out.s] [--loop]

Prints registers / spills from the code object's metadata and the counts the round-6 verdict asked for: v_readlane / v_writelane (SGPR spill
traffic lives in VGPR lanes), s_waitcnt, scratch, DS, VMEM.  --loop: the same counts for the body of the kernel's hottest loop as well — taken
as the longest backward-branch span (the main evaluation loop of k_forward)."""
import collections
import os
import re
import shutil
import subprocess
import sys
import tempfile

LLVM = "/opt/rocm/lib/llvm/bin"


def extract(lib, tmp):
    shutil.copy(lib, os.path.join(tmp, "lib.so"))
    subprocess.run([os.path.join(LLVM, "llvm-objdump"), "--offloading", "lib.so"], cwd=tmp, stdout=subprocess.DEVNULL, stderr=subprocess.DEVNULL)
    return sorted(os.path.join(tmp, f) for f in os.listdir(tmp) if "amdgcn" in f)


def mix(lines):
    c = collections.Counter()
    for l in lines:
        m = re.match(r"\s+([a-z_0-9]+)", l)
        if m:
            c[re.sub(r"_(e32|e64|dpp|sdwa|e64_dpp)$", "", m.group(1))] += 1
    cls = collections.Counter()
    for k, n in c.items():
        cls["VALU" if k.startswith("v_") else "SALU" if k.startswith("s_") else "DS" if k.startswith("ds_") else "SCRATCH" if k.startswith("scratch_") else "VMEM"] += n
    pick = {k: c.get(k, 0) for k in ("v_readlane_b32", "v_writelane_b32", "v_readfirstlane_b32", "v_accvgpr_read_b32", "v_accvgpr_write_b32", "v_mov_b32", "s_waitcnt", "s_nop",
                                     "s_barrier", "s_cbranch_execz", "s_and_saveexec_b64", "v_fma_f32", "v_fmac_f32", "v_mul_f32", "v_add_f32", "v_fma_f64", "v_pk_fma_f32", "v_pk_mul_f32")}
    return sum(c.values()), dict(cls), pick


def main():
    lib, sub = sys.argv[1], sys.argv[2]
    out = next((a for a in sys.argv[3:] if not a.startswith("--")), None)
    tmp = tempfile.mkdtemp(prefix="isa_")
    try:
        for co in extract(os.path.abspath(lib), tmp):
            notes = subprocess.run([os.path.join(LLVM, "llvm-readelf"), "--notes", co], capture_output=True, text=True).stdout
            names = [n for n in re.findall(r"\.name:\s+(\S+)", notes) if sub in n]
            if not names:
                continue
            name = names[0]
            blk = [b for b in re.split(r"\n\s*- \.agpr_count:", notes) if name in b][0]
            meta = {k: re.search(r"\.%s:\s+(\S+)" % k, blk) for k in ("vgpr_count", "sgpr_count", "sgpr_spill_count", "vgpr_spill_count", "private_segment_fixed_size", "group_segment_fixed_size")}
            print(name)
            print("  " + "  ".join("%s=%s" % (k, v.group(1)) for k, v in meta.items() if v), " agpr=" + blk.split()[0])
            dis = subprocess.run([os.path.join(LLVM, "llvm-objdump"), "-d", "--symbolize-operands", co], capture_output=True, text=True).stdout.splitlines()
            a = next(i for i, l in enumerate(dis) if l.endswith("<%s>:" % name))
            b = next((i for i in range(a + 1, len(dis)) if re.match(r"^[0-9a-f]+ <(?!L\d+>)", dis[i])), len(dis))
            body = dis[a:b]
            n, cls, pick = mix(body)
            print("  static instructions: %d  %s" % (n, cls))
            print("  " + "  ".join("%s=%d" % (k.replace("_b32", ""), v) for k, v in pick.items() if v))
            if "--loop" in sys.argv:
                lab = {}
                for i, l in enumerate(body):
                    m = re.match(r"^[0-9a-f]+ <(L\d+)>:", l)
                    if m:
                        lab[m.group(1)] = i
                best = (0, 0, 0)
                for i, l in enumerate(body):
                    m = re.search(r"s_cbranch\w*\s+(L\d+)\b|s_branch\s+(L\d+)\b", l)
                    if m:
                        t = lab.get(m.group(1) or m.group(2))
                        if t is not None and t < i and i - t > best[0]:
                            best = (i - t, t, i)
                if best[0]:
                    n2, cls2, pick2 = mix(body[best[1]:best[2] + 1])
                    print("  longest loop body: %d instructions  %s" % (n2, cls2))
                    print("  " + "  ".join("%s=%d" % (k.replace("_b32", ""), v) for k, v in pick2.items() if v))
            if out:
                open(out, "w").write("\n".join(body))
            return
        raise SystemExit("no kernel matching %r in %s" % (sub, lib))
    finally:
        shutil.rmtree(tmp, ignore_errors=True)


if __name__ == "__main__":
    main()
