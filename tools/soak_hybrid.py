#!/usr/bin/env python3
"""Soak: hybrid cache (W = 400 ring) on the full model shape for more decode steps than the ring is long, then check the
incrementally tracked window sums / accumulators / shadow of every layer against a rebuild from the ring.
    python tools/soak_hybrid.py [--steps 900] [--layers 4]"""
import argparse
import ctypes as C
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), ".."))
import torch  # noqa: E402

from cold_compress_amd import _abi  # noqa: E402
from cold_compress_amd.harness import CONFIGS, GraphedDecoder, ModelArgs, Transformer, prefill, setup_caches  # noqa: E402
from run_configs import HYBRID, Tok  # noqa: E402


def main():
    ap = argparse.ArgumentParser()
    ap.add_argument("--steps", type=int, default=900)
    ap.add_argument("--layers", type=int, default=4)
    ap.add_argument("--prompt", type=int, default=2048)
    a = ap.parse_args()
    dev = torch.device("cuda", 0)
    cfg = dict(CONFIGS["Meta-Llama-3.1-8B-Instruct"])
    cfg["n_layer"] = a.layers
    cfg["block_size"] = a.prompt + a.steps + 64
    torch.manual_seed(7)
    with torch.device("meta"):
        model = Transformer(ModelArgs(**cfg))
    model = model.to_empty(device=dev).to(torch.bfloat16)
    g = torch.Generator(device=dev).manual_seed(7)
    with torch.no_grad():
        for n, p in model.named_parameters():
            p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02, generator=g)
    model.eval()
    kw = dict(max_cache_length=[1.0], cache_bits=None, cache_length_pattern="tile", cache_strategy=["hybrid"],
              cache_strategy_pattern="tile", feed_long_prompts=False, prompt_compression_strategy=["full"], global_tokens=4,
              recent_window=10, history_window_size=1, attn_thresholding=False, min_recovery_frac=0.9, hybrid_strategies=HYBRID)
    setup_caches(model, Tok(), dev, a.prompt + a.steps, kw)
    prompt = torch.randint(0, cfg["vocab_size"], (a.prompt,), generator=torch.Generator().manual_seed(1), dtype=torch.int32).to(dev)
    with torch.no_grad():
        tok, _ = prefill(model, prompt.view(1, -1), torch.arange(a.prompt, device=dev))
        # random weights profile every head as "full": force the reference's policy mix (head h -> policy h % 5) so that
        # the heavy-hitter ring, window, punctuation and special paths all run
        for layer in model.layers:
            kv = layer.attention.kv_cache
            kv.cache_strategies = (torch.arange(kv.n_heads, device=dev) % len(HYBRID)).to(torch.int64).contiguous()
            kv.requires_heavy_hitter = kv.requires_punc = kv.requires_special = True
        dec = GraphedDecoder(model)
        pos = torch.tensor([a.prompt], device=dev, dtype=torch.int32)
        cur = tok.view(1, 1).to(torch.int32)
        for i in range(a.steps):
            cur = dec(model, cur, pos)[0].view(1, 1)
            pos += 1
    torch.cuda.synchronize()
    p = lambda t: C.c_void_p(t.data_ptr())  # noqa: E731
    bad = 0
    for li, layer in enumerate(model.layers):
        kv = layer.attention.kv_cache
        H, S, W = kv.n_heads, kv.max_cache_length, kv.history_window_size
        ws = torch.empty(H * S, device=dev)
        acc = torch.zeros_like(kv.attn_window_acc)
        _abi.call("cc_hh_ring_window_sums", p(kv.attn_history_num), H, S, W, 1, p(ws), p(acc), None)
        torch.cuda.synchronize()
        ok = torch.equal(ws.view(torch.int32), kv.attn_window_sum.reshape(-1).view(torch.int32)) and torch.equal(acc, kv.attn_window_acc)
        print(f"layer {li}: counter {int(kv.attn_counter)} requires_hh {bool(kv.requires_heavy_hitter)} strategies "
              f"{kv.cache_strategies.tolist()} tracked == rebuild: {ok}", flush=True)
        bad += not ok
    from cold_compress_amd.attention_utils import single_launch_status

    kv0 = model.layers[0].attention.kv_cache
    one = int(_abi.lib()["cc_decode_step_hybrid_single_launch"](cfg["n_head"], kv0.n_heads, kv0.max_cache_length, kv0.head_dim, 1))
    st = single_launch_status()
    print(f"single-launch hybrid step for this shape: {bool(one)}; hand-off timeouts: {st}")
    bad += st
    print("SOAK", "FAIL" if bad else "OK")
    sys.exit(1 if bad else 0)


if __name__ == "__main__":
    main()
