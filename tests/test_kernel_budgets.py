"""The register / LDS budgets the two list walkers are tuned for, from the compiler's own resource remarks (no GPU needed: hipcc
cross-compiles gfx950 here).  Round 5 lost a wave per SIMD in the speculative density pass — the list repair code in its verify jobs
took the kernel from 58 to 65 vector registers — and no test said so: the plain variant that bench.py times back to back stayed at 50,
parity does not care, and the step got faster for other reasons.  tools/occ_sweep.sh found it.  This test is the guard:
  k_density_list<1, 0, true>   (the speculative pass of sph_step)  8 waves per SIMD, no scratch
  k_density_list<1, 0, false>  (the slab steps' pass)              8 waves per SIMD, no scratch
  k_force_list<2, 0>           (force + kick + next kick/drift)    7 waves per SIMD, no scratch, an LDS tile that fits 7 times into 160 KB
(DESIGN.md 4.2 "Occupancy sweep": density 25.9 / 28.1 us at 8 / 7 workgroups per CU, force 46.0 / 48.6 at 7 / 6.)
Round 6: the list build (k_build_list, k_rebuild, k_rebuild_slab: one register / LDS budget, the one-launch rebuild's grid is sized by
it) and a guard on the ISA itself — the force pass requests pos_ref with a hand-issued global_load whose s_waitcnt stands in a
later asm statement; between the two the compiler believes the register pair holds a value and could copy or spill it before the
load has landed (round-5 advisor): every path from the load must reach the wait without touching that pair."""
import os
import re
import subprocess

import pytest

from conftest import ROOT

FLAGS = ["--offload-arch=gfx950", "-O3", "-std=c++17", "-fno-slp-vectorize", "-fPIC", "--cuda-device-only", "-Rpass-analysis=kernel-resource-usage"]


@pytest.fixture(scope="module")
def compiled(tmp_path_factory):
    """the kernels compiled ONCE to assembly with the compiler's resource remarks: (remarks on stderr, the .s text)"""
    src = os.path.join(ROOT, "pi-sph-fluid_amd", "csrc", "sph_kernels.hip")
    out = tmp_path_factory.mktemp("isa") / "k.s"
    r = subprocess.run(["/opt/rocm/bin/hipcc"] + FLAGS + ["-I" + os.path.join(ROOT, "include"), "-I" + os.path.dirname(src), "-S", src,
                        "-o", str(out)], capture_output=True, timeout=900)
    assert r.returncode == 0, r.stderr.decode()[-2000:]
    return r.stderr.decode(), out.read_text()


def kernel_resources(remarks):
    out, cur = {}, None
    for line in remarks.splitlines():
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


def test_the_walkers_keep_their_occupancy(compiled):
    res = kernel_resources(compiled[0])

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


def test_the_list_build_keeps_its_budget(compiled):
    """k_build_list / k_rebuild / k_rebuild_slab share build_tile: 6 workgroups of 4 waves per compute unit (registers AND LDS), and the
    one-launch rebuild's grid (rebuild_grid: occupancy x compute units, all resident at once for its grid barriers) follows from it."""
    res = kernel_resources(compiled[0])
    for name in ("k_build_listILi0EEE", "k_rebuildILi0EEE", "k_rebuild_slabILi0EEE"):
        hits = [v for k, v in res.items() if name in k]
        assert len(hits) == 1, name
        k = hits[0]
        assert k["Occupancy [waves/SIMD]"] >= 6, (name, k)
        assert 6 * k["LDS Size [bytes/block]"] <= 160 * 1024, (name, k)
        assert k["ScratchSize [bytes/lane]"] <= 112, (name, k)      # (spills outside the walk, once per tile; k_rebuild: 104 B, the other two 80 B)


def function_text(asm, mangled_part):
    lines = asm.splitlines()
    start = next(i for i, l in enumerate(lines) if re.match(r"^_ZN\S*%s\S*:" % re.escape(mangled_part), l))
    end = next(i for i in range(start, len(lines)) if lines[i].startswith(".Lfunc_end"))
    return lines[start + 1:end]


def touches(code, regs):
    for m in re.finditer(r"\bv(\d+)\b", code):
        if int(m.group(1)) in regs:
            return True
    for m in re.finditer(r"\bv\[(\d+):(\d+)\]", code):
        if any(int(m.group(1)) <= r <= int(m.group(2)) for r in regs):
            return True
    return False


def check_prefetch(lines):
    """Walk the control-flow graph of one function's assembly from every hand-issued `global_load_dwordx2 v[A:B], ..., off` (inline asm:
    between ;;#ASMSTART / ;;#ASMEND) and assert that every path reaches the inline-asm `s_waitcnt vmcnt(0)` before any instruction
    reads or writes vA / vB (the second request of the same pair on the path of a wave that had not asked yet is fine).
    Returns (requests, waits, instructions walked)."""
    ins, labels, in_asm = [], {}, False
    for l in lines:
        t = l.strip()
        if t.startswith(";;#ASMSTART"):
            in_asm = True
            continue
        if t.startswith(";;#ASMEND"):
            in_asm = False
            continue
        m = re.match(r"^(\.LBB\w+):", t)
        if m:
            labels[m.group(1)] = len(ins)
            continue
        code = t.split(";")[0].strip()
        if not code or code.startswith("."):
            continue
        ins.append((code, in_asm))
    asks = [(i, re.match(r"global_load_dwordx2 v\[(\d+):(\d+)\], v\[\d+:\d+\], off$", c)) for i, (c, a) in enumerate(ins)
            if a and c.startswith("global_load_dwordx2")]
    asks = [(i, {int(m.group(1)), int(m.group(2))}) for i, m in asks if m]
    waits = {i for i, (c, a) in enumerate(ins) if a and c == "s_waitcnt vmcnt(0)"}
    walked = 0
    for i0, regs in asks:
        seen, todo = set(), [i0 + 1]
        while todo:
            i = todo.pop()
            if i in seen:
                continue
            assert i < len(ins), "a path from the prefetch runs off the function"
            seen.add(i)
            code, a = ins[i]
            if i in waits:
                continue
            if a and code.startswith("global_load_dwordx2") and touches(code.split(",")[0], regs):
                todo.append(i + 1)
                continue
            assert not touches(code, regs), ("v%s touched between the hand-issued load and its s_waitcnt" % sorted(regs), i, code)
            assert not code.startswith("s_endpgm"), "a path from the prefetch ends without its s_waitcnt"
            assert not code.startswith("s_setpc"), code
            walked += 1
            m = re.match(r"s_c?branch\w*\s+(\.LBB\w+)", code)
            if m:
                todo.append(labels[m.group(1)])
                if code.startswith("s_branch"):
                    continue
            todo.append(i + 1)
    return asks, waits, walked


def test_the_hand_issued_pos_ref_load_is_not_touched_before_its_wait(compiled):
    asks, waits, walked = check_prefetch(function_text(compiled[1], "k_force_listILi2ELi0EEE"))
    assert len(asks) >= 1 and len(waits) >= 1 and walked > 100, (asks, waits, walked)      # (the idiom is still there, and was really walked)
    # the checker itself: a read of the pair between the load and the wait is caught
    bad = ["\t;;#ASMSTART", "\tglobal_load_dwordx2 v[12:13], v[20:21], off", "\t;;#ASMEND", "\tv_mov_b32_e32 v40, v13", "\t;;#ASMSTART",
           "\ts_waitcnt vmcnt(0)", "\t;;#ASMEND", "\ts_endpgm"]
    with pytest.raises(AssertionError):
        check_prefetch(bad)
    ok = [l for l in bad if "v_mov" not in l]
    assert len(check_prefetch(ok)[0]) == 1
