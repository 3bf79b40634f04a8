"""usage: python tools/kernel_sum.py ROUND   (e.g. r06)  -- per workload: the iteration's kernels from profiles/ROUND_kernel_stats_bench_<w>.csv
(rocprofv3 --kernel-trace --stats of `bench.py --workload w`): average duration x launches per iteration, summed = the "rocprof kernel sum" column of BASELINE.md
section 4 (round-over-round claims rest on it, not on one box's wall clock)."""
import csv
import os
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
# kernel-name fragments and launches per steady-state iteration (error iterations: every tenth)
SPEC = {
    "c2": [("k_factor_product_x3", 2), ("k_mu64_update32<false", 1), ("k_mu64_update32<true", 1)],
    "c5": [("k_factor_product_x3", 2), ("k_panel_update64_lds_f32", 2), ("k_mu64_reduce_scale_all_x3", 1), ("k_reduce_partials", 1), ("k_row_dot_part", 0.1), ("k_trace_small", 0.1), ("k_copy_small", 0.1)],
    "c5g": [("k_factor_product_x3", 2), ("k_panel_update64_lds_f32", 2), ("k_mu64_reduce_scale_all_x3", 1), ("k_row_dot_part", 0.1), ("k_trace_small", 0.1), ("k_copy_small", 0.1)],
    "c4": [("k_factor_product_bf16_r2", 2), ("k_panel_update_rows_mu", 1), ("k_panel_update_wide_f32", 1), ("k_trace_small", 0.1), ("k_copy_small", 0.1)],
    "c3": [("k_kl_fused<float, 2, false", 1.9), ("k_kl_fused<float, 2, true", 0.1), ("k_kl_update", 2), ("k_kl_sums", 2), ("k_normalize_panel_v2", None), ("k_compact_partials", None), ("k_gram_wide_x3", 0.2)],
    "c2f64": [("k_factor_product_f64", 2), ("k_panel_update64_f64", 2), ("k_gram_f64", None), ("k_gram_reduce_sym_f64", None), ("k_normalize_panel_v2", None), ("k_compact_partials", None), ("k_reduce_partials", None), ("k_trace_small", 0.1), ("k_copy_small", 0.1)],
    "example": [("k_factor_product_f64", 2), ("k_panel_update_wide_f64", 2), ("k_gram_f64", None), ("k_gram_reduce_sym_f64", None), ("k_smooth_panel", None), ("k_normalize_panel_v2", None), ("k_compact_partials", None), ("k_trace_small", 0.1), ("k_copy_small", 0.1)],
}


def main():
    rnd = sys.argv[1] if len(sys.argv) > 1 else "r06"
    for w, spec in SPEC.items():
        path = os.path.join(ROOT, "profiles", f"{rnd}_kernel_stats_bench_{w}.csv")
        if not os.path.exists(path):
            continue
        rows = list(csv.DictReader(open(path)))
        # launches per iteration marked None: present only in older rounds' sequences -- counted by their share of the product launches
        prod = next((float(r["Calls"]) for r in rows if spec[0][0] in r["Name"]), None)
        total_prod = sum(float(r["Calls"]) for r in rows if spec[0][0] in r["Name"])
        iters = total_prod / spec[0][1] if spec[0][1] else 1.0
        total, parts = 0.0, []
        for frag, per in spec:
            match = [r for r in rows if frag in r["Name"]]
            if not match:
                continue
            calls = sum(float(r["Calls"]) for r in match)
            avg = sum(float(r["Calls"]) * float(r["AverageNs"]) for r in match) / calls / 1e3
            n = per if per is not None else round(calls / iters, 2)
            total += avg * n
            parts.append(f"{frag.split('<')[0]} {n:g} x {avg:.2f}")
        print(f"{rnd} {w:8s} rocprof kernel sum {total:8.1f} us per iteration   ({'; '.join(parts)})")


if __name__ == "__main__":
    main()
