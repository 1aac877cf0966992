"""The register / LDS budgets the two list walkers are tuned for, from the compiler's own resource remarks (no GPU needed: hipcc
cross-compiles gfx950 here).  Round 5 lost a wave per SIMD in the speculative density pass — the list repair code in its verify jobs
took the kernel from 58 to 65 vector registers — and no test said so: the plain variant that bench.py times back to back stayed at 50,
parity does not care, and the step got faster for other reasons.  tools/occ_sweep.sh found it.  This test is the guard:
  k_density_list<1, 0, true>   (the speculative pass of sph_step)  8 waves per SIMD, no scratch
  k_density_list<1, 0, false>  (the slab steps' pass)              8 waves per SIMD, no scratch
  k_force_list<2, 0>           (force + kick + next kick/drift)    7 waves per SIMD, no scratch, an LDS tile that fits 7 times into 160 KB
(DESIGN.md 4.2 "Occupancy sweep": density 25.9 / 28.1 us at 8 / 7 workgroups per CU, force 46.0 / 48.6 at 7 / 6.)"""
import os
import re
import subprocess

from conftest import ROOT

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]


def kernel_resources(tmp_path):
    src = os.path.join(ROOT, "pi-sph-fluid_amd", "csrc", "sph_kernels.hip")
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-c", src,
                        "-o", str(tmp_path / "k.co")], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    out, cur = {}, None
    for line in r.stderr.decode().splitlines():
        m = re.search(r"remark: (.*?) *\[-Rpass", line)
        if not m:
            continue
        t = m.group(1).strip()
        if t.startswith("Function Name:"):
            cur = t.split(":", 1)[1].strip()      # (the mangled name)
            out[cur] = {}
        elif cur and ":" in t:
            k, v = t.split(":", 1)
            if v.strip().isdigit():
                out[cur][k.strip()] = int(v)
    return out


def test_the_walkers_keep_their_occupancy(tmp_path):
    res = kernel_resources(tmp_path)

    mangled = {"k_density_list<1, 0, true>": "k_density_listILi1ELi0ELb1EEE", "k_density_list<1, 0, false>": "k_density_listILi1ELi0ELb0EEE",
               "k_force_list<2, 0>": "k_force_listILi2ELi0EEE"}

    def of(name):
        hits = [v for k, v in res.items() if mangled[name] in k]
        assert len(hits) == 1, (name, sorted(res)[:40])
        return hits[0]
    for name, waves in (("k_density_list<1, 0, true>", 8), ("k_density_list<1, 0, false>", 8), ("k_force_list<2, 0>", 7)):
        k = of(name)
        assert k["Occupancy [waves/SIMD]"] >= waves, (name, k)
        assert k["ScratchSize [bytes/lane]"] == 0 and k["VGPRs Spill"] == 0, (name, k)
    # 160 KB of LDS per compute unit: the force tile seven times, the density tile eight times
    assert 7 * of("k_force_list<2, 0>")["LDS Size [bytes/block]"] <= 160 * 1024
    assert 8 * of("k_density_list<1, 0, true>")["LDS Size [bytes/block]"] <= 160 * 1024
