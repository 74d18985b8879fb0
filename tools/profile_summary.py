#!/usr/bin/env python3
"""Condense rocprofv3 output directories into the summaries committed under profiles/.

    python tools/profile_summary.py --round r01 --stats gpurun_out/prof_r1 \
        --fetch gpurun_out/pmc_fetch --write gpurun_out/pmc_write

* --stats : directory of `rocprofv3 --kernel-trace --stats --output-format csv -d DIR -- python bench.py ...`
* --fetch / --write : directories of the separate `--pmc FETCH_SIZE` / `--pmc WRITE_SIZE` passes
  (each with `--kernel-trace` only, as MI355X_MICROARCH.md's HBM section prescribes).

Writes profiles/<round>_bench_kernel_stats.csv (copy of the newest *_kernel_stats.csv), profiles/<round>_summary.txt
and profiles/<round>_k1_traffic.json (HBM bytes per launch of the main-pass top-k kernel; bench.py reads it for
`roofline.traffic`).  gfx950 correction: FETCH_SIZE under-reports wide coalesced reads by 2x; the l2norm_rows_kernel
row (compulsory 1536 MB read, 768 MB written at 1 M x 384) in the same pass is the calibration.
"""
from __future__ import annotations

import argparse
import csv
import glob
import json
import os
import shutil
from collections import defaultdict

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def newest(pattern: str):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getmtime) if files else None


def largest(pattern: str):
    files = glob.glob(pattern, recursive=True)
    return max(files, key=os.path.getsize) if files else None


def pmc_per_kernel(directory: str, counter: str):
    """{kernel name: (mean counter value per launch, launches)} from the newest counter csv in the directory."""
    f = newest(os.path.join(directory, "**", "*_counter_collection.csv"))
    if f is None:
        return {}
    acc = defaultdict(list)
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter:
                acc[r["Kernel_Name"]].append(float(r["Counter_Value"]))
    return {k: (sum(v) / len(v), len(v)) for k, v in acc.items()}


def pmc_main_pass_per_search(directory: str, counter: str, first: int):
    """The counter of the first `first` main-pass launches, in dispatch order: the benchmark steps come first (two launches
    per search: phase A, phase B); the Q = 256 probes bench.py runs afterwards (one launch each) are left out."""
    f = newest(os.path.join(directory, "**", "*_counter_collection.csv"))
    vals = []
    with open(f, newline="") as fh:
        for r in csv.DictReader(fh):
            if r["Counter_Name"] == counter and MAIN_PASS in r["Kernel_Name"]:
                vals.append((int(r["Dispatch_Id"]), float(r["Counter_Value"])))
    return [v for _, v in sorted(vals)][:first]


MAIN_PASS = "cos_topk_partial_kernelILi384ELi8ELi1ELi16ELb0ELb1ELb0ELb0ELb1ELi3E"   # (mangled: the half-precision signature does not demangle)


def main() -> None:
    ap = argparse.ArgumentParser()
    ap.add_argument("--round", default="r01")
    ap.add_argument("--stats", required=True)
    ap.add_argument("--fetch")
    ap.add_argument("--write")
    ap.add_argument("--queries", type=int, default=4096)
    ap.add_argument("--rows", type=int, default=1000000)
    ap.add_argument("--d", type=int, default=384)
    ap.add_argument("--k", type=int, default=10)
    ap.add_argument("--cmd", default="python bench.py --steps 10 --warmup 2 --no-cpu-baseline")
    ap.add_argument("--searches", type=int, default=4, help="searches (steps + warm-up) of the PMC command")
    ap.add_argument("--layers", type=int, default=6)
    ap.add_argument("--preset", default="all-MiniLM-L6-v2")
    ap.add_argument("--cmd-pmc", default="python bench.py --steps 3 --warmup 1 --no-cpu-baseline --no-verify")
    ap.add_argument("--launches-per-search", type=int, default=2,
                    help="main-pass launches of one tsim_cosine_topk call (2 since round 2: phase A + phase B)")
    a = ap.parse_args()
    out_dir = os.path.join(ROOT, "profiles")
    lines = [f"kernel stats: rocprofv3 --kernel-trace --stats -- {a.cmd}", ""]

    stats = newest(os.path.join(a.stats, "**", "*_kernel_stats.csv"))
    if stats is None:
        raise SystemExit(f"no *_kernel_stats.csv under {a.stats}")
    shutil.copyfile(stats, os.path.join(out_dir, f"{a.round}_bench_kernel_stats.csv"))
    lines.append(f"{'calls':>6} {'avg_us':>10} {'total_ms':>9} {'%':>6}  kernel")
    with open(stats, newline="") as fh:
        for r in csv.DictReader(fh):
            lines.append(f"{int(r['Calls']):6d} {float(r['AverageNs']) / 1e3:10.1f} {float(r['TotalDurationNs']) / 1e6:9.2f} "
                         f"{float(r['Percentage']):6.2f}  {r['Name'][:100]}")

    traffic = None
    if a.fetch and a.write:
        fe, wr = pmc_per_kernel(a.fetch, "FETCH_SIZE"), pmc_per_kernel(a.write, "WRITE_SIZE")
        lines += ["", "PMC (separate passes, --pmc FETCH_SIZE / --pmc WRITE_SIZE, per launch averages; values in KiB as reported;",
                  "gfx950 correction per MI355X_MICROARCH.md HBM section: FETCH_SIZE x2 for wide coalesced reads)"]
        for name in fe:
            if "tsim::" not in name and "4tsim" not in name:   # (half-precision signatures come out mangled: _ZN4tsim...)
                continue
            f_kib, n = fe[name]
            w_kib = wr.get(name, (0.0, 0))[0]
            lines.append(f"  FETCH {f_kib:12.0f} KiB (x2 = {2 * f_kib * 1024 / 1e6:9.1f} MB)  WRITE {w_kib:10.0f} KiB "
                         f"({w_kib * 1024 / 1e6:8.1f} MB)  n={n}  {name[:90]}")
            if MAIN_PASS in name:      # the main pass (list kernel, PAIR schedule): per search = its launches of the step's grid
                fv = pmc_main_pass_per_search(a.fetch, "FETCH_SIZE", a.searches * a.launches_per_search)
                wv = pmc_main_pass_per_search(a.write, "WRITE_SIZE", a.searches * a.launches_per_search)
                searches = len(fv) // a.launches_per_search
                f_kib, w_kib = sum(fv) / max(searches, 1), sum(wv) / max(searches, 1)
                lines.append(f"  main pass per search ({a.launches_per_search} launches, {searches} searches): FETCH x2 = "
                             f"{2 * f_kib * 1024 / 1e6:.1f} MB, WRITE = {w_kib * 1024 / 1e6:.1f} MB")
                traffic = {
                    "round": int(a.round.lstrip("r")), "gpu": "MI355X (gfx950)",
                    "command": "python bench.py --steps 3 --warmup 1 --no-cpu-baseline",
                    "per": f"tsim_cosine_topk call = {a.launches_per_search} launches of this kernel (phase A rows [0, 131072), phase B the rest)",
                    "workload": {"queries_per_step": a.queries, "corpus_rows_per_gpu": a.rows, "d": a.d, "k": a.k},
                    "kernel": "cos_topk_partial_kernel<384,8,1,16,MAXONLY=false,PAIR=true,COLLECT=false,PP=false,M16=true,NST=3>",
                    "fetch_size_kib": int(f_kib), "write_size_kib": int(w_kib),
                    "hbm_bytes_per_launch": int(2 * f_kib * 1024 + w_kib * 1024),
                    "correction": "FETCH_SIZE x2 (gfx950 reports half the bytes of wide coalesced reads, MI355X_MICROARCH.md "
                                  "HBM section; calibrated in the same run on l2norm_rows_kernel: 1536 MB read, 768 MB written); "
                                  "WRITE_SIZE exact",
                    "passes": ["rocprofv3 --pmc FETCH_SIZE --kernel-trace", "rocprofv3 --pmc WRITE_SIZE --kernel-trace"],
                }
    # encoder: HBM bytes per layer = sum over the layer's kernels (projections, attention, LayerNorm GEMMs and their remainder
    # launches) of FETCH x 2 + WRITE, over all forwards of the PMC command, divided by forwards x layers
    if a.fetch and a.write:
        layer_kernels = ("gemm_xres", "attention_kernel", "ln_rows_gemm", "ln_tail_gemm", "gemm_bf16_kernel", "ffn_fused", "gemm_pp", "res_ln_rows")
        tot_f = sum(v * n for k, (v, n) in fe.items() if any(t in k for t in layer_kernels))
        tot_w = sum(v * n for k, (v, n) in wr.items() if any(t in k for t in layer_kernels))
        per_layer = (2 * tot_f + tot_w) * 1024 / (a.searches * a.layers)
        lines += ["", f"encoder: FETCH x 2 + WRITE over the layer kernels of {a.searches} forwards x {a.layers} layers = "
                      f"{per_layer / 1e6:.1f} MB per layer"]
        for k in sorted(fe):
            if any(t in k for t in layer_kernels):
                fk, n = fe[k]
                wk = wr.get(k, (0.0, 0))[0]
                lines.append(f"   per launch {(2 * fk + wk) * 1024 / 1e6:8.1f} MB  x {n:3d}  {k[:100]}")
        with open(os.path.join(out_dir, f"{a.round}_encoder_traffic.json"), "w") as fh:
            json.dump({"round": int(a.round.lstrip("r")), "gpu": "MI355X (gfx950)", "command": a.cmd_pmc,
                       "workload": {"preset": a.preset, "sentences_per_step": a.queries},
                       "hbm_bytes_per_layer": int(per_layer), "layers": a.layers, "forwards": a.searches,
                       "definition": "sum over the layer's kernels (QKV / FFN1 projections, attention, LayerNorm GEMMs with their "
                                     "remainder launches) of FETCH_SIZE x 2 + WRITE_SIZE, all forwards of the command, / (forwards x layers)",
                       "correction": "FETCH_SIZE x2 (gfx950, MI355X_MICROARCH.md HBM section); WRITE_SIZE exact"}, fh, indent=1)
            fh.write("\n")
    with open(os.path.join(out_dir, f"{a.round}_summary.txt"), "w") as fh:
        fh.write("\n".join(lines) + "\n")
    if traffic is not None:
        with open(os.path.join(out_dir, f"{a.round}_k1_traffic.json"), "w") as fh:
            json.dump(traffic, fh, indent=1)
            fh.write("\n")
    print("\n".join(lines))


if __name__ == "__main__":
    main()
