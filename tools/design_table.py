#!/usr/bin/env python3
"""The measurement table of DESIGN.md section 8 from the committed records (profiles/r06_bench_detail.json: the default
`python bench.py` run; profiles/pmc_counters.json: the rocprofv3 passes of tools/profile_round.sh) -- so that the table is
typed by a program, not by hand.  usage: design_table.py [round tag, default r06]"""
import json
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    tag = sys.argv[1] if len(sys.argv) > 1 else "r06"
    d = json.load(open(os.path.join(ROOT, "profiles", f"{tag}_bench_detail.json")))
    p = json.load(open(os.path.join(ROOT, "profiles", "pmc_counters.json")))
    rows = [("**config 3** local_opt H=10 (headline)", d, "cfg3"), ("config 2 finite_horizon H=10", d["config2"], "cfg2"),
            ("config 4, per-GPU share of 8", d["config4_share8"], "cfg4_share8"), ("config 5, per-GPU share of 8", d["config5_share8"], "cfg5_share8"),
            ("config 4 whole on ONE GPU", d["config4_whole"], "cfg4"), ("config 5 whole on ONE GPU", d["config5_whole"], "cfg5"),
            ("the reference's shape, H=5, pop 9 x 3", d["reference_h5"], "reference_h5"),
            ("validation planner H=6, n_iter 200, K=6", d["reference_h6_extra"], "reference_h6_extra")]
    print("| workload | episodes / launch | kernel | kernel ms: HIP events / rocprofv3 steady avg (calls) | step ms | episodes/s | fp32-vector frac | HBM counters / algorithmic | CPU oracle |")
    print("|---|---|---|---|---|---|---|---|---|")
    for name, b, key in rows:
        r, c = b["roofline"], p[key]
        ll = r["launch"]
        E = c["episodes_per_launch"]
        algo = r.get("algorithmic_bytes_per_episode", 48) * E
        traffic = (c["fetch_kib"] + c["write_kib"]) * 1024
        kern = f"`{r['kernel_symbol'].split('ocd::')[1].split('(')[0]}` {ll['mapping']}" + (f" S={ll['chunk']}" if ll["chunk"] else "") + \
               (" latency build" if ll["build_wavefronts_per_simd"] == 1 else f" {ll['build_wavefronts_per_simd']} per SIMD")
        cpu = b["cpu_baseline"]
        print(f"| {name} | {E} | {kern} | {r['kernel_ms']:.3f} / {c['kernel_steady_avg_us'] / 1e3:.3f} ({c['kernel_calls']}) | {b['ms_per_step']:.3f} | "
              f"{b['value']:.4g} | {100 * r['binding_frac']:.2f} % | {traffic / algo:.1f}x ({traffic / 1e6:.2f} MB) | {cpu['value']:.0f}/s on {cpu['cores']} threads |")
    x = d["reference_h5_x28"]
    print(f"| 28 such runs in lockstep, {x['launch_groups']} groups, {x.get('host_threads', '?')} host threads | {x['episodes_per_generation']} | one launch of all: {x['kernel_ms']:.3f} ms | "
          f"generation {x['cma_generation_ms']:.3f} wall / {x['cma_generation_native_timers_ms']:.3f} native | | {x['value']:.4g} | | | {x['cpu_baseline']['value']:.0f}/s |")
    c1 = d["config1"]
    print(f"| config 1 (plumbing: 3 inits, designer weights) | 3 | kernel {c1['kernel_ms']:.3f} ms | eval_weights {c1['eval_weights_ms']:.3f} ms; world.step() {c1['world_step_ms']:.3f} ms | | | | | {c1['cpu_baseline']['value']:.0f}/s on 1 thread |")
    print()
    print("cma_generation_ms", d["cma_generation_ms"], "host split", d["cma"].get("host_split_ms"))
    print("collective", d["collective"])
    print("predicted_strong_scaling", json.dumps(d["predicted_strong_scaling"]))
    print("line bytes", os.path.getsize(os.path.join(ROOT, "profiles", f"{tag}_bench.json")))


if __name__ == "__main__":
    main()
