"""One table of current numbers per kernel from a measurement set of tools/r4_final_profile.sh:

    python tools/r4_table.py profiles r4        # reads profiles/r4_bench_kernel_stats.csv, _hbm_traffic.json, _pmc_sq.txt, _bench_line.json (+ the 8-bit twins)

Kernel launch forms (template arguments: tile shape, operand type, tail split) are grouped under one row each.  Algorithmic work per launch comes
from the shapes of the workload (config 2: S = 4289; the 8-bit image runs config 5's S = 4354 in the bench's fp8 leg but THIS profile is the
`--precision int8 ...` run of config 2's shape, S = 4289).
"""
import collections
import csv
import json
import os
import re
import sys

root, tag = sys.argv[1], sys.argv[2]
S, T, D, M, H = 4289, 193, 3072, 12288, 24
BF16_PEAK, I8_PEAK, HBM_PEAK = 2500.0, 5000.0, 8.0


def stats(fn):
    rows = collections.OrderedDict()
    with open(fn) as fh:
        for r in csv.DictReader(fh):
            rows[r["Name"]] = (int(r["Calls"]), float(r["TotalDurationNs"]), float(r["AverageNs"]))
    return rows


def pmc(fn):
    out, cur = {}, None
    if not os.path.exists(fn):
        return out
    for line in open(fn):
        if not line.startswith(" "):
            cur = line.split("  launches")[0].strip()
            out[cur] = {}
        elif cur:
            k, v = line.split()[:2]
            out[cur][k] = float(v)
    return out


def group(name):
    n = name.replace(" ", "")
    m = re.search(r"td_gemm_bf16_nt_kernel<(\d+),(\d+),(true|false),(true|false),(true|false)", n)
    if m:
        wm, wn, conv, fp8, i8 = m.groups()
        kind = "conv" if conv == "true" else "e4m3" if fp8 == "true" else "int8" if i8 == "true" else "bf16"
        return f"GEMM {32 * int(wm)}x{64 * int(wn)} {kind}"
    for key, lab in (("td_attn_fwd_d128_streamk", "attention bf16 (stream-K)"), ("td_attn_fwd_d128_fp8", "attention e4m3"), ("td_attn_fp8_pack", "attention e4m3 pack pass"),
                     ("td_attn_fwd_d128_lean", "attention bf16 (plain grid)"), ("td_norm_rows_kernel", "LayerNorm + modulate"), ("td_qk_norm_rope", "QK-RMSNorm + RoPE"),
                     ("td_quant_rows", "int8 / e4m3 quantisation pass"), ("td_q8_scales", "history scales"), ("td_euler", "Euler step"), ("td_col_amax", "smoothing calibration (column maxima)"),
                     ("td_ext_cols", "smoothing (replicated weight columns)")):
        if key in n:
            return lab
    return None


def flops(label, calls):
    """Algorithmic FLOPs per launch: attention 4 S^2 H 128; the block Linears of one image by the tile shape that runs them (256x256:
    q|k|v, to_out, ff.net.0 of the 19 double blocks and proj_mlp|q|k|v of the 38 single blocks; 288x192: ff.net.2 and proj_out), averaged over the
    group's launches (the few embedder / modulation launches in the group carry ~0.1 % of its FLOPs)."""
    if label.startswith("attention") and "pack" not in label:
        return 4.0 * S * S * H * 128
    if label.startswith("GEMM 256x256") and "conv" not in label:
        return 28 * (19 * 2.0 * S * D * (3 * D + D + M) + 38 * 2.0 * S * D * (3 * D + M)) / calls
    if label.startswith("GEMM 288x192"):
        return 28 * (19 * 2.0 * S * M * D + 38 * 2.0 * S * (D + M) * D) / calls
    return None


def table(stats_fn, pmc_fn, traffic_fn, peak, title, images=1):
    st = stats(stats_fn)
    pm = pmc(pmc_fn)
    tr = json.load(open(traffic_fn))["kernels"] if traffic_fn and os.path.exists(traffic_fn) else {}
    g = collections.OrderedDict()
    tot = sum(v[1] for k, v in st.items() if "fill_normal" not in k and "rocclr" not in k)
    for name, (calls, total, avg) in st.items():
        lab = group(name)
        if lab is None:
            continue
        e = g.setdefault(lab, {"calls": 0, "ns": 0.0, "mfma_busy": 0.0, "wave": 0.0, "valu": 0.0, "mfma": 0.0, "bytes": 0.0, "bytes_n": 0, "wait": 0.0})
        e["calls"] += calls
        e["ns"] += total
        c = pm.get(name, {})
        e["mfma_busy"] += c.get("SQ_VALU_MFMA_BUSY_CYCLES", 0.0) * calls
        e["wave"] += c.get("SQ_BUSY_CYCLES", 0.0) * calls
        e["valu"] += c.get("SQ_INSTS_VALU", 0.0) * calls
        e["mfma"] += c.get("SQ_INSTS_MFMA", 0.0) * calls
        e["wait"] += c.get("SQ_WAIT_ANY", 0.0) * calls
        t = tr.get(name)
        if t:
            e["bytes"] += t["hbm_bytes_per_launch"] * t["launches"]
            e["bytes_n"] += t["launches"]
    tot /= images
    for e in g.values():
        e["calls"] //= images
        e["ns"] /= images
    print(f"\n**{title}** (`{os.path.basename(stats_fn)}`; per image of 28 steps{'' if images == 1 else f', mean of the {images} images of the run'}; {tot / 1e6:.0f} ms of kernels)\n")
    print("| kernel | launches / image | avg us | ms / image | share | rate | of peak | VALU per MFMA | HBM bytes / launch (PMC) |")
    print("|---|---|---|---|---|---|---|---|---|")
    for lab, e in g.items():
        if e["ns"] / tot < 0.002:
            continue
        avg_us = e["ns"] / e["calls"] / 1e3
        rate = frac = ""
        fl = flops(lab, e["calls"]) if e["calls"] >= 500 or lab.startswith("attention") else None      # (a handful of launches: the calibration forward's bf16 GEMMs in the 8-bit run)
        if fl:
            r = fl / (avg_us * 1e-6) / 1e12
            rate, frac = f"{r:.0f} TFLOP/s", f"{r / peak:.3f}"
        vm = f"{e['valu'] / e['mfma']:.2f}" if e["mfma"] > 0 else ""
        by = f"{e['bytes'] / e['bytes_n'] / 1e6:.0f} MB" if e["bytes_n"] else ""
        if by and not fl:
            tb = e["bytes"] / e["bytes_n"] / (avg_us * 1e-6) / 1e12
            rate, frac = f"{tb:.2f} TB/s", f"{tb / HBM_PEAK:.2f}"
        print(f"| {lab} | {e['calls']} | {avg_us:.1f} | {e['ns'] / 1e6:.1f} | {100 * e['ns'] / tot:.1f} % | {rate} | {frac} | {vm} | {by} |")


p = lambda s: os.path.join(root, f"{tag}_{s}")
table(p("bench_kernel_stats.csv"), p("pmc_sq.txt"), p("hbm_traffic.json"), BF16_PEAK, "bf16 image (the headline's arithmetic)")
if os.path.exists(p("int8_smooth_attn8_kernel_stats.csv")):
    # (two images in that run: the warm-up image carries the calibration forward -- bf16 GEMMs + column maxima -- and the timed one does not)
    table(p("int8_smooth_attn8_kernel_stats.csv"), p("pmc_sq_int8_smooth_attn8.txt"), None, I8_PEAK, "8-bit image (int8 Linears: smoothing + history scales; e4m3 attention)", images=2)
d = json.load(open(p("bench_line.json")))
print(f"\nheadline: {d['value']:.4f} images/s ({d['ms_per_step']:.1f} ms per step of {d['config']['images_per_rank_per_step']} images), one at a time {d['one_image_in_flight']['value']:.4f}; "
      f"dominant GEMM {d['roofline']['achieved']:.0f} TFLOP/s = {d['roofline']['frac']:.3f} (events, {d['roofline']['avg_launch_us']:.1f} us), attention {d['attention_roofline']['frac']:.3f}; "
      f"whole step {d['whole_step_tflops_per_gpu']:.0f} TFLOP/s = {d['whole_step_tflops_per_gpu'] / BF16_PEAK:.3f}; 8-bit `fp8.value` {d['fp8']['value']:.4f} ({d['fp8']['policy']}, inside the bar on every fixture: {d['fp8']['inside_1e-2_bar']}); "
      f"cpu_baseline {d['cpu_baseline']['value']:.5f} images/s on {d['cpu_baseline']['cores']} cores")
