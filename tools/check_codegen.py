"""Static check of what DESIGN.md section 6 ("Round 4") found decisive for the one-launch kernels, on a `hipcc -S` dump
of their translation unit (csrc/Makefile: -DESKF_TU_FUSED -mllvm -disable-machine-licm):
  * no FLAT memory instructions (a pointer loaded from the kernarg segment is generic unless it is laundered into the
    global address space -- `globalise` --; FLAT accesses count on the LDS counter as well and serialise with the exchange),
  * the register budget of four wavefronts per SIMD (<= 128 VGPRs) with at most a handful of spilled registers,
  * no scratch access inside the rounds loop of the run kernel (loop depth >= 2): what is spilled is touched once per step.

usage: check_codegen.py <asm file>   -> prints one line per kernel, exit code 1 on a violation
"""
import re
import sys


def kernels(path):
    """name -> (body lines, metadata dict) for every kernel of the dump"""
    lines = open(path).read().split("\n")
    out = {}
    i = 0
    while i < len(lines):
        m = re.match(r"^(_Z\w+):\s", lines[i] + " ")
        if m and i > 0 and "@function" in "".join(lines[max(0, i - 3):i]):
            name = m.group(1)
            j = i + 1
            while j < len(lines) and not lines[j].startswith(".Lfunc_end"):
                j += 1
            out[name] = (lines[i + 1:j], {})
            i = j
        i += 1
    name = None
    for ln in lines:
        m = re.match(r"\s*\.amdhsa_kernel\s+(\S+)", ln)
        if m:
            name = m.group(1)
        m = re.match(r"\s*\.amdhsa_(next_free_vgpr|private_segment_fixed_size|accum_offset)\s+(\d+)", ln)
        if m and name in out:
            out[name][1][m.group(1)] = int(m.group(2))
    return out


def check(path, key, max_vgpr=128, max_scratch=128, rounds_depth=None):
    """-> list of (kernel, n_flat, vgprs, scratch bytes, scratch instructions inside the rounds loop, ok); the rounds loop is
    the innermost one: depth 2 in the run kernel (inside the time loop), depth 1 in the one-launch step"""
    if rounds_depth is None:
        rounds_depth = 2 if "run_kernel" in key else 1
    res = []
    for name, (body, meta) in kernels(path).items():
        if key not in name:
            continue
        n_flat = deep = 0
        depth = 0
        for ln in body:
            s = ln.strip()
            if re.match(r"\.LBB\d+_\d+:", s):
                m = re.search(r"Depth=(\d+)", s)
                depth = int(m.group(1)) if m else 0
                continue
            op = s.split()[0] if s else ""
            if op.startswith("flat_"):
                n_flat += 1
            if op.startswith("scratch_") and depth >= rounds_depth:
                deep += 1
        # (gfx950 has a unified register file: next_free_vgpr counts architectural + accumulation registers)
        vg = meta.get("accum_offset", meta.get("next_free_vgpr", 0))
        sc = meta.get("private_segment_fixed_size", 0)
        ok = n_flat == 0 and 0 < vg <= max_vgpr and sc <= max_scratch and deep == 0
        res.append((name, n_flat, vg, sc, deep, ok))
    return res


if __name__ == "__main__":
    bad = 0
    for key in ("eskf_step_kernel", "eskf_run_kernel"):
        for r in check(sys.argv[1], key):
            print("%s: flat %d, vgprs %d, scratch %d B, scratch accesses inside the rounds %d -> %s" % (*r[:5], "ok" if r[5] else "VIOLATION"))
            bad += not r[5]
    sys.exit(1 if bad else 0)
