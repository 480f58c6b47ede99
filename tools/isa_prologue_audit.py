"""Audit the single-launch step kernels' prologues in the device ISA: every `s_waitcnt vmcnt(N)` between a kernel's first LDS-DMA
load (`buffer_load ... lds`) and the first `s_barrier` behind it.  A small N there means the wave sits out its K tile before it has
requested the rest (q, the V tile): how the r4 stalls of the l2 / recent_global / full / random steps were found.

  hipcc --offload-arch=gfx950 -O3 -std=c++17 -ffp-contract=off -mllvm -amdgpu-mfma-vgpr-form=1 -S --cuda-device-only \
        -o /tmp/dec.s cold_compress_amd/csrc/cc_attn_decode.hip
  python tools/isa_prologue_audit.py /tmp/dec.s
"""
import re
import sys


def main(path):
    name, state, dma, rows = None, 0, 0, []
    out = {}
    for ln, line in enumerate(open(path), 1):
        m = re.match(r"^(_ZN\S*decode_attn_split_mfma_kernel\S*):", line)
        if m:
            name, state, dma, rows = m.group(1), 0, 0, []
            continue
        if name is None:
            continue
        t = line.strip()
        if t.startswith(".Lfunc_end"):
            if state:
                out[name] = rows
            name = None
        elif " lds" in t and t.startswith("buffer_load") and state < 2:
            state, dma = 1, dma + 1
        elif state == 1 and t.startswith("s_barrier"):
            state = 2
        elif state == 1 and t.startswith("s_waitcnt") and "vmcnt" in t:
            n = int(re.search(r"vmcnt\((\d+)\)", t).group(1))
            rows.append((ln, n, dma))
    for k, rows in out.items():
        args = re.search(r"kernelI(\S+?)EEv", k).group(1)
        bad = [r for r in rows if r[1] < r[2]]  # waits for at least one DMA load issued so far
        print(f"{args}: {len(rows)} vmcnt waits in the prologue; waiting on DMA loads: {[(l, n) for l, n, _ in bad]}")


if __name__ == "__main__":
    main(sys.argv[1])
