"""Audit (r6, CC_V_PRELOAD): does anything WAIT for memory between a step kernel's entry and its first K request?

For every decode_attn_split_mfma_kernel instantiation whose tile arrives by LDS-DMA (`buffer_load ... lds`), lists the `s_waitcnt`s
that name lgkmcnt (scalar / kernel-argument loads) or vmcnt between the kernel's real entry — behind the 256-byte compatibility
prologue of the preloaded arguments — and the first DMA load, and the number of instructions in between.  The product wants none:
the operands of the first requests are preloaded SGPRs.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -mllvm -amdgpu-kernarg-preload-count=14 \
        -Icold_compress_amd/csrc -Iinclude -S --cuda-device-only -o /tmp/dec.s cold_compress_amd/csrc/cc_attn_decode.hip
  python tools/isa_first_request_audit.py /tmp/dec.s
"""
import os
import re
import subprocess
import sys


def main(path):
    name, body, out = None, [], []
    for line in open(path):
        m = re.match(r"^(_ZN\S*decode_attn_split_mfma_kernel\S*):", line)
        if m:
            name, body = m.group(1), []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            out.append((name, body))
            name = None
        elif t and not t.startswith(";"):
            body.append(t)
    names = subprocess.run(["c++filt"], input="\n".join(n for n, _ in out), capture_output=True, text=True).stdout.split("\n")
    bad = 0
    for (n, body), dn in zip(out, names):
        if not any(" lds" in t and t.startswith("buffer_load") for t in body):
            continue
        start = 0
        for i, t in enumerate(body):  # the compatibility prologue ends at the first `.p2align 8` (preload builds only)
            if t.startswith(".p2align") and "8" in t:
                start = i + 1
                break
        waits, n_ins = [], 0
        for t in body[start:]:
            if " lds" in t and t.startswith("buffer_load"):
                break
            if t.endswith(":") or t.startswith("."):
                continue
            n_ins += 1
            if t.startswith("s_waitcnt") and ("lgkmcnt" in t or "vmcnt" in t):
                waits.append(t)
        # (the loop over key rows beyond 256 entries holds a vmcnt(0): never taken at the product's sizes)
        real = [w for w in waits if "lgkmcnt" in w]
        # EARLY (CC_V_EARLYARGS): the block of scalar loads from the kernarg pointer s[0:1] at the kernel's entry is issued in assembly and
        # waited for by hand behind the K request — until that wait NOTHING may read or overwrite the registers they are filling
        pend, touched, k = set(), [], start
        while k < len(body) and not body[k].startswith("s_load_dword"):
            if body[k].startswith("s_waitcnt") and "lgkmcnt" in body[k] and real:  # (FULL: the entry stamps' own wait, ahead of the block)
                real = real[1:]
            k += 1
        while k < len(body) and body[k].startswith("s_load_dword") and "s[0:1]" in body[k]:
            m = re.match(r"s_load_dword(x\d+)?\s+s\[?(\d+)(?::(\d+))?\]?", body[k])
            lo = int(m.group(2)); hi = int(m.group(3)) if m.group(3) else lo
            pend |= set(range(lo, hi + 1))
            k += 1
        seen_dma = False
        for t in body[k:]:
            if " lds" in t and t.startswith("buffer_load"):
                seen_dma = True
            if seen_dma and t.startswith("s_waitcnt") and "lgkmcnt(0)" in t:
                break
            for m in re.finditer(r"\bs\[(\d+):(\d+)\]|\bs(\d+)\b", t):
                regs = range(int(m.group(1)), int(m.group(2)) + 1) if m.group(1) else [int(m.group(3))]
                if pend & set(regs):
                    touched.append(t)
                    break
        if os.environ.get("CC_AUDIT_VERBOSE"):
            print(f"   early block: {len(pend)} scalar registers pending; instructions touching them before their wait: {len(touched)}")
        if touched and len(pend) > 14:
            bad += 1
            print(f"PENDING REGISTERS TOUCHED before their wait in <{dn[:120]}>: {touched[:4]}")
        args = re.search(r"decode_attn_split_mfma_kernel<(.*?)>\(", dn)
        tag = "EARLY WAIT" if real else "ok"
        bad += 1 if real else 0
        print(f"{tag:10s} {n_ins:4d} instr to the first K request  <{args.group(1) if args else dn}>  {real}")
    print(f"{bad} instantiation(s) wait for scalar memory in front of their first K request")


if __name__ == "__main__":
    main(sys.argv[1])
