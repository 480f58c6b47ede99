#!/usr/bin/env python3
"""One layer of the Llama-3-70B shape (dim 8192, 64 / 8 heads): the decode loop's fused path (hand-written GEMVs + single-launch step)
against the same model with `fuse_gemv = False` (hipBLASLt Linears, the fused step still) and with the three-call cache path — which
piece disagrees at this shape?  Prints the largest probability difference per token and variant."""
import os
import sys

sys.path.insert(0, os.path.join(os.path.dirname(os.path.abspath(__file__)), "..", ".."))
import torch  # noqa: E402

from cold_compress_amd.harness import CONFIGS, ModelArgs, Transformer, decode_one_token, prefill, setup_caches  # noqa: E402


def main():
    dev = torch.device("cuda", 0)
    cfg = dict(CONFIGS[sys.argv[1] if len(sys.argv) > 1 else "Llama-3-70B-shape"])
    cfg.update(n_layer=1, block_size=4096, vocab_size=4096)
    torch.manual_seed(77)
    model = Transformer(ModelArgs(**cfg)).to(torch.bfloat16).eval()
    with torch.no_grad():
        g = torch.Generator().manual_seed(77)
        for n, p in model.named_parameters():
            p.fill_(1.0) if "norm" in n else p.normal_(0.0, 0.02, generator=g)
    model = model.to(dev)
    kw = dict(max_cache_length=[3488.0], cache_bits=None, cache_length_pattern="tile", cache_strategy=["heavy_hitter"],
              cache_strategy_pattern="tile", feed_long_prompts=False, prompt_compression_strategy=["heavy_hitter"], global_tokens=4,
              recent_window=10, history_window_size=1, attn_thresholding=False, min_recovery_frac=0.9)
    L = 3600
    prompt = torch.randint(0, cfg["vocab_size"], (L,), generator=torch.Generator().manual_seed(3), dtype=torch.int32).to(dev)
    runs = {}
    for name, fg, fs in (("fused", True, True), ("no_gemv", False, True), ("three_call", False, False), ("gemv_three_call", True, False)):
        for l in model.layers:
            l.fuse_gemv = fg
            l.attention.fuse_decode_step = fs
        setup_caches(model, None, dev, L + 100, dict(kw))
        with torch.no_grad():
            tok, probs = prefill(model, prompt.view(1, -1), torch.arange(L, device=dev))
            pos = torch.tensor([L], dtype=torch.int32, device=dev)
            plist, toks = [probs.float().clone()], [int(tok)]
            cur = tok.view(1, 1).to(torch.int32)
            for t in range(6):
                nt, pr = decode_one_token(model, cur, pos)
                plist.append(pr.float().clone())
                toks.append(int(nt))
                cur = torch.tensor([runs["fused"][1][len(toks) - 1] if "fused" in runs else int(nt)], device=dev).view(1, 1).to(torch.int32)
                pos += 1
        torch.cuda.synchronize()
        runs[name] = (plist, toks)
    ref = runs["three_call"][0]
    for name in runs:
        d = [float((a - b).abs().max() / b.abs().max()) for a, b in zip(runs[name][0], ref)]
        print(name, "tokens", runs[name][1], "max|dp|/max p vs three_call:", [round(x, 4) for x in d], flush=True)


if __name__ == "__main__":
    main()
