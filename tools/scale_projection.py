#!/usr/bin/env python3
"""PROJECTION (not a measurement) of bench.py --gpus N for N in {1, 2, 4, 8}: what the first real scaling curve should be compared
with.  Inputs that ARE measured on one MI355X (profiles/r04_bench.json, tools/ab_step.py at the per-rank
shapes) and the ASSUMPTIONS for what has never run here (anything that crosses xGMI), all written into the output.

    python tools/scale_projection.py profiles/r04_bench.json '{"8": 8.41, "4": 7.89, "2": 7.30, "1": 7.05}' > profiles/r04_scale_projection.json
"""
import json
import sys


def main():
    bench = json.load(open(sys.argv[1]))
    step_us = {int(k): float(v) for k, v in json.loads(sys.argv[2]).items()}  # kv heads per rank -> measured layer-step us (S = 4096)
    n_layer = bench["config"]["n_layer"]
    wt = bench["whole_token"]
    w_bytes = wt["weight_bytes"]
    lm_head = 128256 * 4096 * 2  # replicated on every rank (tp.py:171-176), like the final norm
    layer_w = w_bytes - lm_head
    t1 = bench["ms_per_step"] * 1e-3
    gemv1 = t1 - n_layer * step_us[8] * 1e-6
    # one-GPU calibration of the dense part: bytes / BW + launches x overhead, BW fixed at 6.0 TB/s (the GEMVs' measured streaming
    # rate, DESIGN / r01_gemv_vs_hipblaslt), overhead solved from the measured token
    bw = 6.0e12
    launches = 5 * n_layer + 1
    ovh = (gemv1 - w_bytes / bw) / launches
    assume = {
        "gemv_stream_bw_bytes_per_s": bw,
        "per_launch_overhead_s_solved_from_the_measured_token": ovh,
        "allreduces_per_token": 2 * n_layer,
        "allreduce_bytes": 2 * 4096,
        "oneshot_allreduce_us": 5.0,  # ASSUMED: one launch boundary (~2.1 us measured between dependent kernels here) + one flag round trip over xGMI (2-3 us, never measured)
        "rccl_allreduce_us": 20.0,    # ASSUMED: RCCL small-message latency inside a captured graph (never measured here)
        "lm_head_replicated_bytes": lm_head,
        "what_is_not_modelled": "load imbalance between ranks, the one-shot transport failing its self-test (then RCCL), clock / power differences of an 8-GPU node, "
                                "smaller GEMVs streaming below 6 TB/s (w2 at TP = 8 is 14.7 MB per rank: ramp-bound)",
    }
    rows = []
    for n in (1, 2, 4, 8):
        h = 8 // n
        dense = (layer_w / n + lm_head) / bw + launches * ovh
        steps = n_layer * step_us[h] * 1e-6
        row = {"n_gpus": n, "kv_heads_per_rank": h, "per_rank_weight_bytes": int(layer_w / n + lm_head), "dense_ms": round(dense * 1e3, 3),
               "layer_steps_ms": round(steps * 1e3, 3), "layer_step_us_measured_one_gpu": step_us[h]}
        for name, us in (("oneshot", assume["oneshot_allreduce_us"]), ("rccl", assume["rccl_allreduce_us"])):
            ar = 0.0 if n == 1 else assume["allreduces_per_token"] * us * 1e-6
            tok = dense + steps + ar
            row[f"tokens_per_s_{name}"] = round(1.0 / tok, 1)
            row[f"ms_per_token_{name}"] = round(tok * 1e3, 3)
        if n == 1:
            row["measured_tokens_per_s"] = bench["value"]
        rows.append(row)
    print(json.dumps({"WHAT_THIS_IS": "a PROJECTION from one-GPU measurements and stated assumptions — no multi-GPU run has happened (no 2-GPU or "
                                      "8-GPU box was available to the builder in any round; the driver runs the N = 1, 2, 4, 8 bench itself); nothing here is a result",
                      "workload": bench["config"]["workload"], "scaling": "strong (TP = N, the same token stream)",
                      "measured_inputs": {"one_gpu_ms_per_token": bench["ms_per_step"], "one_gpu_tokens_per_s": bench["value"],
                                          "layer_step_us_by_kv_heads_per_rank": step_us, "weight_bytes": w_bytes},
                      "assumptions": assume, "projection": rows}, indent=1))


if __name__ == "__main__":
    main()
