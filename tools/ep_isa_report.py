"""ISA report of the long-horizon kernel: per role (the code between the role's entry branch and the next), the number of
vector / scalar / LDS / scratch / lane-spill instructions, and the same for the covariance wavefronts' predict loop.
usage: ep_isa_report.py [f64|f32] [NC]   (compiles dvi_ekf_amd/csrc/eskf_epochs.hip to assembly first)"""
import os, re, subprocess, sys
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
dt = "d" if (len(sys.argv) < 2 or sys.argv[1] == "f64") else "f"
nc = sys.argv[2] if len(sys.argv) > 2 else "2"
asm = "/tmp/ep.s"
if not os.environ.get("EP_NO_BUILD"):
    subprocess.check_call(["/opt/rocm/bin/hipcc", "-O3", "-std=c++17", "-fPIC", "--offload-arch=gfx950", "-mllvm", "-disable-machine-licm", "-S",
                           "--cuda-device-only", "-o", asm, os.path.join(root, "dvi_ekf_amd/csrc/eskf_epochs.hip")] + sys.argv[3:], stderr=subprocess.DEVNULL)
pp = os.environ.get("EP_PP", "1")
name = "_ZN4eskf18eskf_epochs_kernelI%sLi%sELi%sEEEvNS_10EpochsArgsENS_9DevParamsIT_EE" % (dt, nc, pp)
lines, on = [], False
for l in open(asm):
    if l.startswith(name + ":"):
        on = True
    if on:
        lines.append(l.rstrip("\n"))
        if "s_endpgm" in l:
            break
def count(seg):
    c = {"valu": 0, "dpp": 0, "salu": 0, "lds": 0, "scratch": 0, "lane_spill": 0, "vmem": 0, "nop": 0}
    for l in seg:
        t = l.strip()
        if t.startswith("v_readlane") or t.startswith("v_writelane"): c["lane_spill"] += 1
        elif t.startswith("v_"): c["valu"] += 1; c["dpp"] += "_dpp" in t
        elif t.startswith("s_nop"): c["nop"] += 1
        elif t.startswith("s_"): c["salu"] += 1
        elif t.startswith("ds_"): c["lds"] += 1
        elif t.startswith("scratch_"): c["scratch"] += 1
        elif t.startswith("global_") or t.startswith("flat_") or t.startswith("buffer_"): c["vmem"] += 1
    return c
print(len(lines), "lines", count(lines))
# loops: by the compiler's "Loop Header" comments
hdr = [i for i, l in enumerate(lines) if "Loop Header: Depth=" in l or "Inner Loop Header: Depth=" in l]
# the predict loop of the covariance wavefronts: the innermost loop that contains v_permlane16_swap and v_fmac_*_dpp with ds_write
labels = {}
for i, l in enumerate(lines):
    m = re.match(r"^(\.LBB\d+_\d+):", l)
    if m: labels[m.group(1)] = i
loops = []
for i, l in enumerate(lines):
    m = re.search(r"s_c?branch\S*\s+(\.LBB\d+_\d+)", l)
    if m and m.group(1) in labels and labels[m.group(1)] < i:
        loops.append((labels[m.group(1)], i))
for a, b in sorted(loops, key=lambda ab: ab[1] - ab[0]):
    seg = lines[a:b + 1]
    c = count(seg)
    if c["valu"] > 60:
        tag = "covariance" if c["dpp"] > 50 else ""
        print("loop lines %d-%d (%d)" % (a, b, b - a), c, tag)
