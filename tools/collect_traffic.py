"""Assemble profiles/<round>/traffic_by_config.json -- what bench.py quotes as `roofline.traffic` for every single-GPU
configuration -- from the per-configuration rocprofv3 sets tools/profile_config.sh left under gpurun_out/, and copy each
set's summaries (kernel_stats.csv, pmc_summary.txt, traffic.json, bench_under_rocprof.json) into profiles/<round>/.
usage: collect_traffic.py <round> <prefix> key=tag [key=tag ...]     e.g.  collect_traffic.py r03 final cfg3=r3f_cfg3 cfg2=r3f_cfg2"""
import json
import os
import shutil
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))


def main():
    rnd, prefix = sys.argv[1], sys.argv[2]
    dst = os.path.join(ROOT, "profiles", rnd)
    os.makedirs(dst, exist_ok=True)
    out_path = os.path.join(dst, "traffic_by_config.json")
    out = json.load(open(out_path)) if os.path.exists(out_path) else {}
    for kv in sys.argv[3:]:
        key, tag = kv.split("=")
        src = os.path.join(ROOT, "gpurun_out", f"prof_{tag}")
        for f in ("kernel_stats.csv", "pmc_summary.txt", "traffic.json", "bench_under_rocprof.json"):
            if os.path.exists(os.path.join(src, f)):
                shutil.copy(os.path.join(src, f), os.path.join(dst, f"{prefix}_{key}_{f}"))
        t = json.load(open(os.path.join(src, "traffic.json")))
        keep = ("bytes", "fetch_size_kib", "write_size_kib", "avg_ns_under_stats", "clock_ghz", "valu_insts",
                "valu_busy_cycles_per_simd", "valu_busy_frac_under_profiler", "valu_floor_ms")
        out[key] = {"files": f"{prefix}_{key}_*", "command": t.get("workload", ""), "tag": t.get("tag", tag),
                    "kernels": {k: {f: v[f] for f in keep if f in v} for k, v in t["kernels"].items()
                                if not k.startswith(("__amd", "at::", "void at::"))}}
    json.dump(out, open(out_path, "w"), indent=1, sort_keys=True)
    print(out_path, sorted(out))


if __name__ == "__main__":
    main()
