"""Turns the rocprofv3 outputs of one profiling run into the summaries committed under profiles/.

    python tools/profile_summarise.py <tag> <stats_dir> <pmc_fetch_dir> <pmc_write_dir> [<pmc_sq_dir>]

  <stats_dir>      rocprofv3 --kernel-trace --stats --output-format csv -d <stats_dir> -- python3 bench.py ...
  <pmc_*_dir>      rocprofv3 --pmc FETCH_SIZE | WRITE_SIZE | SQ_... --output-format csv -d <dir> -- python3 bench.py ...
                   (separate passes, counters only: MI355X_MICROARCH.md "rocprofv3 PMC slots")
Writes profiles/<tag>_kernel_stats.csv (our kernels only) and profiles/<tag>_pmc_traffic.json with, per kernel,
FETCH_SIZE / WRITE_SIZE averages per launch and hbm_bytes_corrected = 2*FETCH_SIZE + WRITE_SIZE (gfx950 counts a 128-byte
read request as 64 bytes: MI355X_MICROARCH.md, HBM section), plus mfma_util when the SQ pass is given."""
import csv, glob, json, os, re, sys
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
OURS = ("ppo_train", "rollout_h2", "gae_kernel", "clip_adam", "slab_reduce", "adv_stats", "policy_", "synth_rware", "mlp_forward",
        "pack_w1", "gru_scan", "rec_dense", "rec_xty", "seq_", "rec_step", "t32_convert", "coop", "rec_gather", "rec_pack",
        "norm_act", "colsum", "im2col", "col2im", "flatten_kernel")


def short(name: str) -> str:
    name = re.sub(r"\(anonymous namespace\)::", "", name)
    name = re.sub(r"^void ", "", name)
    return re.sub(r"\(.*$", "", name)


def find(d, pat):
    fs = glob.glob(os.path.join(d, "**", pat), recursive=True)
    return fs[0] if fs else None


def counters(d):
    out = defaultdict(lambda: defaultdict(list))
    f = find(d, "*counter_collection.csv")
    if not f:
        return out
    for row in csv.DictReader(open(f)):
        name = short(row["Kernel_Name"])
        out[name][row["Counter_Name"]].append(float(row["Counter_Value"]))
        try:  # the same kernel at different grids (actor: 256 blocks, the once-per-env critic: 32) also per grid
            blocks = int(row["Grid_Size"]) // max(1, int(row["Workgroup_Size"]))
            out[f"{name} grid={blocks}"][row["Counter_Name"]].append(float(row["Counter_Value"]))
        except (KeyError, ValueError):
            pass
    return out


def main():
    tag, stats_dir, rd, wd = sys.argv[1:5]
    sq = sys.argv[5] if len(sys.argv) > 5 else None
    f = find(stats_dir, "*kernel_stats.csv")
    rows = [r for r in csv.DictReader(open(f))]
    keep = [r for r in rows if any(k in r["Name"] for k in OURS)]
    with open(os.path.join(ROOT, "profiles", f"{tag}_kernel_stats.csv"), "w", newline="") as g:
        w = csv.DictWriter(g, fieldnames=list(rows[0].keys()))
        w.writeheader()
        for r in keep:
            r = dict(r)
            r["Name"] = short(r["Name"])
            w.writerow(r)
    cr, cw = counters(rd), counters(wd)
    cs = counters(sq) if sq else {}
    kernels = {}
    for name in sorted(set(cr) | set(cw)):
        if not any(k in name for k in OURS):
            continue
        fs, ws = cr.get(name, {}).get("FETCH_SIZE", []), cw.get(name, {}).get("WRITE_SIZE", [])
        if not fs or not ws:
            continue
        fk, wk = sum(fs) / len(fs), sum(ws) / len(ws)
        e = {"launches": len(fs), "FETCH_SIZE_KB": round(fk, 1), "WRITE_SIZE_KB": round(wk, 1),
             "hbm_bytes_uncorrected": int((fk + wk) * 1024), "hbm_bytes_corrected": int((2 * fk + wk) * 1024)}
        if name in cs and "SQ_VALU_MFMA_BUSY_CYCLES" in cs[name] and "GRBM_GUI_ACTIVE" in cs[name]:
            busy = sum(cs[name]["SQ_VALU_MFMA_BUSY_CYCLES"]) / len(cs[name]["SQ_VALU_MFMA_BUSY_CYCLES"])
            act = sum(cs[name]["GRBM_GUI_ACTIVE"]) / len(cs[name]["GRBM_GUI_ACTIVE"])
            e["mfma_util"] = round(busy / (act / 8 * 1024), 4)  # SIMD-cycles of the launch: 8 XCDs summed, 1024 SIMDs
            e["gpu_cycles_per_launch"] = int(act / 8)
        kernels[name] = e
    doc = ("rocprofv3 --pmc FETCH_SIZE / --pmc WRITE_SIZE" + (" / --pmc SQ_VALU_MFMA_BUSY_CYCLES SQ_BUSY_CYCLES GRBM_GUI_ACTIVE" if sq else "")
           + " (separate passes, counters only) on `python3 bench.py --steps 1 --warmup 2 --no-cpu-baseline --no-kernel-timers`; "
           "averages per launch. gfx950: FETCH_SIZE tallies a 128-B read request as 64 B, so hbm_bytes_corrected = 2*FETCH_SIZE + WRITE_SIZE "
           "(MI355X_MICROARCH.md, HBM section); exact for wide streaming reads, an upper bound otherwise.")
    json.dump({"_doc": doc, "kernels": kernels}, open(os.path.join(ROOT, "profiles", f"{tag}_pmc_traffic.json"), "w"), indent=1)
    print(f"wrote profiles/{tag}_kernel_stats.csv ({len(keep)} kernels) and profiles/{tag}_pmc_traffic.json ({len(kernels)} kernels)")


if __name__ == "__main__":
    main()
