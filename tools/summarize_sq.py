"""profiles/<tag>_sq_counters.txt + profiles/<tag>_sq_resident.txt (the lines tools/pmc.sh / tools/pmc_cmd.sh print) ->
profiles/valu_counters.json: per regime of the bench line the vector-ALU issue share of the kernels that serve it,

    valu_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32 shader engines)

weighted by the kernels' busy cycles x launches where a regime is several kernels, with the counters it comes from
(per launch).  bench.py quotes these constants next to its flop fractions (`valu_issue`), the way it quotes the PMC traffic.
usage: summarize_sq.py <tag>"""
import ast, json, os, re, sys

tag = sys.argv[1]
root = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def parse(path):
    out = []
    if not os.path.exists(path):
        return out
    for line in open(path):
        m = re.search(r"^(\[[^\]]*\] )?(\S+) (.*?) (\{'SQ_.*?\}) (\{.*?\}) launches (\d+)", line)
        if not m:
            m2 = re.search(r"^(\[[^\]]*\] )?(\S+) (.*?) (\{'SQ_.*?\}) launches (\d+)", line)
            if not m2:
                continue
            pre, t, name, ctr, n = m2.groups()
        else:
            pre, t, name, ctr, _, n = m.groups()
        out.append({"tag": (pre or "").strip() + t, "kernel": name.strip(), "counters": ast.literal_eval(ctr), "launches": int(n)})
    return out


rows = parse(os.path.join(root, "profiles", f"{tag}_sq_counters.txt")) + parse(os.path.join(root, "profiles", f"{tag}_sq_resident.txt"))


def pick(tag_part, *name_parts):
    return [r for r in rows if tag_part in r["tag"] and any(p in r["kernel"] for p in name_parts)]


def summarize(sel):
    if not sel:
        return None
    act = sum(r["counters"]["SQ_ACTIVE_INST_VALU"] * r["launches"] for r in sel)
    busy = sum(r["counters"]["SQ_BUSY_CYCLES"] * r["launches"] for r in sel)
    return {"valu_busy": round(act / (8.0 * busy), 3),
            "kernels": [{"kernel": r["kernel"], "launches": r["launches"], "valu_busy": round(r["counters"]["SQ_ACTIVE_INST_VALU"] / (8.0 * r["counters"]["SQ_BUSY_CYCLES"]), 3),
                         "SQ_INSTS_VALU_per_wave": round(r["counters"]["SQ_INSTS_VALU"] / max(r["counters"]["SQ_WAVES"], 1.0), 1),
                         "SQ_ACTIVE_INST_VALU": r["counters"]["SQ_ACTIVE_INST_VALU"], "SQ_INSTS_VALU": r["counters"]["SQ_INSTS_VALU"],
                         "SQ_BUSY_CYCLES": r["counters"]["SQ_BUSY_CYCLES"], "SQ_WAVES": r["counters"]["SQ_WAVES"]} for r in sel]}


EPOCH = ("eskf_nominal_kernel", "eskf_jac_kernel", "eskf_cov_rows_kernel", "eskf_pre_kernel", "eskf_epochs_kernel")
res = {
    "source": f"profiles/{tag}_sq_counters.txt, profiles/{tag}_sq_resident.txt (rocprofv3 --pmc, counters in their own pass; tools/prof_round.sh)",
    "formula": "valu_busy = 4 SQ_ACTIVE_INST_VALU / (1024 SIMDs x SQ_BUSY_CYCLES / 32); several kernels: weighted by busy cycles x launches",
    "headline_pairs_fp64_B65536": summarize([r for r in rows if r["tag"] == f"{tag}_sq" and "eskf_run_kernel<double" in r["kernel"]]),
    "epochs_10to1_fp64": summarize(pick("sq_epoch10", *EPOCH)),
    "pairs_fp64_B8192": summarize(pick("sq_small", "eskf_run_kernel<double, 1")),
    "config5_f32_B1024": summarize([r for r in pick("sq_c5", "eskf_epochs_kernel<float, 2") if "EPOCHS=0" not in r["tag"]]),
    "config5_f64_B1024": summarize([r for r in pick("sq_c5", "eskf_epochs_kernel<double, 2") if "EPOCHS=0" not in r["tag"]]),
    "config5_f32_B8192": summarize([r for r in pick("sq_c5", "eskf_epochs_kernel<float, 8") if "EPOCHS=0" not in r["tag"]]),
    "config5_f64_B8192": summarize([r for r in rows if r["tag"].endswith("sq_c5") and "<double" in r["kernel"] and any(p in r["kernel"] for p in EPOCH[:3])]),
    "config5_f32_B1024_epoch_by_epoch": summarize([r for r in rows if "EPOCHS=0" in r["tag"] and "<float" in r["kernel"] and any(p in r["kernel"] for p in EPOCH[:3])]),
}
json.dump(res, open(os.path.join(root, "profiles", "valu_counters.json"), "w"), indent=1)
print(json.dumps({k: (v["valu_busy"] if isinstance(v, dict) else v) for k, v in res.items() if k not in ("source", "formula")}, indent=1))
