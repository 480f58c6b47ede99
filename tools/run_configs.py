#!/usr/bin/env python3
"""End-to-end runs of the SURVEY §8(d) configurations on the full Llama-3-8B shape (bf16, random weights): 8k/16k-token
prompt prefilled through the HIP path, then hipGraph decode; prints decode tokens/s and the cache statistics per config.
    python tools/run_configs.py [--steps 32]"""
import argparse
import json
import os
import sys
import time

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cold_compress_amd.harness import CONFIGS, GraphedDecoder, ModelArgs, Transformer, prefill, setup_caches  # noqa: E402

HYBRID = [{"strategy": "special"}, {"strategy": "special_punc"}, {"strategy": "special_punc_heavy_hitter", "heavy_hitter_frac": 0.3},
          {"strategy": "special_punc_window", "recent_window": 0.3}, {"strategy": "full"}]


class Tok:
    def special_ids(self):
        return [[1], [2, 3]]

    def punctuation_ids(self):
        return [5, 6, 7, 11, 13]


def run(name, cache, prompt_len, steps, dev):
    cfg = dict(CONFIGS["Meta-Llama-3.1-8B-Instruct"])
    cfg["block_size"] = prompt_len + 2048 + 64
    torch.manual_seed(1234)
    with torch.device("meta"):
        model = Transformer(ModelArgs(**cfg))
    model = model.to_empty(device=dev).to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(1234)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02, generator=g)
    model.eval()
    kw = dict(max_cache_length=[4096.0], cache_bits=None, cache_length_pattern="tile", cache_strategy=["heavy_hitter"],
              cache_strategy_pattern="tile", feed_long_prompts=False, prompt_compression_strategy=["heavy_hitter"], global_tokens=4,
              recent_window=10, history_window_size=1, attn_thresholding=False, min_recovery_frac=0.9, hybrid_strategies=HYBRID)
    kw.update(cache)
    prompt = torch.randint(0, cfg["vocab_size"], (prompt_len,), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(dev)
    with torch.no_grad():  # one untimed prefill: library heuristics, allocator growth and clocks settle before the timed one
        setup_caches(model, Tok(), dev, prompt_len + 2048, dict(kw))
        prefill(model, prompt.view(1, -1), torch.arange(prompt_len, device=dev))
        torch.cuda.synchronize()
    ck = setup_caches(model, Tok(), dev, prompt_len + 2048, kw)
    with torch.no_grad():
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        tok, _ = prefill(model, prompt.view(1, -1), torch.arange(prompt_len, device=dev))
        torch.cuda.synchronize()
        tp = time.perf_counter() - t0
        pos = torch.tensor([prompt_len], dtype=torch.int32, device=dev)
        cur = tok.view(1, 1).to(torch.int32)
        dec = GraphedDecoder(model)
        for _ in range(4):
            cur = dec(model, cur, pos)[0].view(1, 1)
            pos += 1
        torch.cuda.synchronize()
        t0 = time.perf_counter()
        for _ in range(steps):
            cur = dec(model, cur, pos)[0].view(1, 1)
            pos += 1
        torch.cuda.synchronize()
        dt = time.perf_counter() - t0
    st = model.get_cache_stats(prompt_len, steps + 4)
    lens = sorted(set(int(x) for x in ck["max_cache_length"]))
    print(json.dumps({"config": name, "prompt": prompt_len, "cache_lengths": lens if len(lens) < 6 else [lens[0], "...", lens[-1]],
                      "prefill_s": round(tp, 2), "decode_tok_s": round(steps / dt, 1), "ms_per_token": round(dt / steps * 1e3, 3),
                      "compression_ratio_avg": round(st["compression_ratio_avg"], 4), "cache_memory_gb": round(st["cache_memory_gb"], 3)}),
          flush=True)
    del model, dec
    torch.cuda.empty_cache()


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=32)
    ap.add_argument("--only", default=None)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cases = [
        ("C2 heavy_hitter 0.25", dict(max_cache_length=[0.25]), 8192),
        ("C3 heavy_hitter 4096", dict(), 8192),
        ("C3 l2 4096", dict(cache_strategy=["l2"], prompt_compression_strategy=["l2"]), 8192),
        ("C3 random 4096", dict(cache_strategy=["random"], prompt_compression_strategy=["random"]), 8192),
        ("recent_global 4096", dict(cache_strategy=["recent_global"], prompt_compression_strategy=["recent_global"]), 8192),
        ("C4 hybrid 1.0", dict(cache_strategy=["hybrid"], prompt_compression_strategy=["full"], max_cache_length=[1.0]), 16384),
        ("C4 heavy_hitter pyramid 1024", dict(max_cache_length=[1024.0], cache_length_pattern="pyramid"), 16384),
        ("heavy_hitter 4096 cache_bits=8", dict(cache_bits=8), 8192),
        ("heavy_hitter 4096 cache_bits=8 fused", dict(cache_bits=8, cache_quant_mode="fused"), 8192),
    ]
    for name, cache, pl in cases:
        if a.only and a.only not in name:
            continue
        try:
            run(name, cache, pl, a.steps, dev)
        except Exception as e:  # keep going: this is a survey of configurations
            print(json.dumps({"config": name, "error": f"{type(e).__name__}: {e}"[:300]}), flush=True)
            torch.cuda.empty_cache()


if __name__ == "__main__":
    main()
