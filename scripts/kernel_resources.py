#!/usr/bin/env python3
"""Register / scratch / LDS use of every kernel of a csrc/*.hip file, from hipcc's
-Rpass-analysis=kernel-resource-usage (cross-compiles: no GPU needed).

    python3 scripts/kernel_resources.py bm25 [-DNAME ...]
"""
import os
import re
import subprocess
import sys

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
sys.path.insert(0, ROOT)


def resources(base, defines=()):
    import triple_hybrid_rag_amd as T
    src = os.path.join(T._build.CSRC, base + ".hip")
    cmd = ["/opt/rocm/bin/hipcc"] + T._build.HIPCC_FLAGS + list(defines) + \
          ["-Rpass-analysis=kernel-resource-usage", "-c", src, "-o", "/dev/null"]
    err = subprocess.run(cmd, capture_output=True, text=True).stderr
    out, cur = [], None
    for line in err.splitlines():
        m = re.search(r"remark:\s+(.*?) \[-Rpass", line)
        if not m:
            continue
        text = m.group(1).strip()
        if text.startswith("Function Name:"):
            name = text.split(":", 1)[1].strip()
            demangled = subprocess.run(["c++filt", name], capture_output=True,
                                       text=True).stdout.strip()
            cur = {"kernel": re.sub(r"\(.*", "", demangled).replace("void thr::", "")}
            out.append(cur)
        elif cur is not None and ":" in text:
            k, v = text.split(":", 1)
            cur[k.strip()] = v.strip()
    return out


if __name__ == "__main__":
    for r in resources(sys.argv[1], sys.argv[2:]):
        print(f"{r['kernel'][:70]:70s} VGPR {r.get('VGPRs'):>4s} AGPR {r.get('AGPRs'):>3s} "
              f"spill {r.get('VGPR Spill', r.get('VGPRs Spill', '?')):>3s} scratch {r.get('ScratchSize [bytes/lane]'):>4s} "
              f"occ {r.get('Occupancy [waves/SIMD]'):>2s} LDS {r.get('LDS Size [bytes/block]')}")
