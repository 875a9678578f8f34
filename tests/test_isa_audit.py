"""Static audit of the hand-scheduled int8 Gram kernels (csrc/kernels_gram_i8p.h, csrc/kernels_gram_i8.h), CPU only: hipcc cross-compiles the
release instantiations to gfx950 assembly and the test reads it.

Why.  Both kernels issue their global loads / LDS-DMA / ds_reads from inline asm and count vmcnt / lgkmcnt by hand: the compiler does not know
that a fragment register is still waiting for its data.  That is only sound while the compiler emits NOTHING of its own that touches those registers
between an asm load and its wait -- no v_mov copy, no accumulator move, no spill -- and nothing that would disturb the hand-counted queues (no VMEM /
LDS instruction of its own inside the loop).  The guide's rule ("audit after every edit", cdna_hip_programming.md) as a test: inside every loop that
holds MFMAs only scalar bookkeeping and VALU address adds may appear outside the asm statements; no spills, no scratch."""
import os
import re
import subprocess
import tempfile
from collections import Counter

import pytest

ROOT = os.path.dirname(os.path.dirname(os.path.abspath(__file__)))
CSRC = os.path.join(ROOT, "plspm-python_amd", "csrc")
HIPCC = "/opt/rocm/bin/hipcc"

SOURCE = r'''
#include <hip/hip_runtime.h>
#include <cstdint>
#include "%(csrc)s/philox.h"
#include "%(csrc)s/kernels_gram_i8.h"
#include "%(csrc)s/kernels_gram_i8p.h"
#define P_ARGS const uint4*, const uint4*, int, int, int, int, int, const int*, const double*, int, long, double*, long, int
#define R_ARGS const uint4*, const uint4*, int, int, int, int, int, const int*, const int*, const double*, int, long, double*, long, int
template __global__ void gram_i8p_kernel<6, 5, 64, false>(P_ARGS);
template __global__ void gram_i8p_kernel<6, 5, 64, true>(P_ARGS);
template __global__ void gram_i8p_kernel<7, 4, 64, false>(P_ARGS);
template __global__ void gram_i8p_kernel<7, 4, 64, true>(P_ARGS);
#define PP_ARGS const uint4*, const uint4*, int, int, int, int, int, const int*, const double*, int, long, double*, long, int, GramI8PPCtl*
template __global__ void gram_i8pp_kernel<6, 5, 64, true>(PP_ARGS);
template __global__ void gram_i8pp_kernel<7, 4, 64, true>(PP_ARGS);
template __global__ void gram_i8_kernel<6, 4, 3, 16, 20, false>(R_ARGS);
template __global__ void gram_i8_kernel<7, 4, 803, 16, 16, false>(R_ARGS);
template __global__ void gram_i8_kernel<5, 4, 3, 16, 16, false>(R_ARGS);
'''
ALLOWED = re.compile(r"^(s_|v_lshl_add_u64|v_add_u32|v_add_co_u32|v_addc_co_u32|v_lshl_add_u32|v_lshlrev_b32|v_and_b32|v_or_b32|v_add3_u32|v_cndmask_b32|v_readfirstlane_b32)")


@pytest.fixture(scope="module")
def assembly():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc in this environment")
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "audit.hip")
        with open(src, "w") as fh:
            fh.write(SOURCE % {"csrc": CSRC})
        out = os.path.join(d, "audit.s")
        subprocess.check_call([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-S", "--cuda-device-only", "-Wno-unused-value", "-o", out, src])
        return open(out).read()


def kernels(asm):
    for m in re.finditer(r"^(_Z\d+gram_i8p{0,2}_kernel\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def mfma_loops(body):
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"\.LBB\w+:", l.strip())}
    for i, l in enumerate(body):
        mm = re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = body[labels[mm.group(1)]:i + 1]
            # (innermost loops only: the persistent kernel's tile loop holds the k-step loop AND the epilogue, which is the compiler's to schedule)
            inner = any(re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", x) and labels.get(re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", x).group(1), 1 << 30) >= labels[mm.group(1)]
                        and labels.get(re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", x).group(1), 1 << 30) < i and k < len(seg) - 1
                        and labels.get(re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", x).group(1), 1 << 30) <= labels[mm.group(1)] + k
                        for k, x in enumerate(seg[:-1]))
            if any("v_mfma" in x for x in seg) and not inner:
                yield seg


def test_no_compiler_instruction_touches_the_hand_counted_queues(assembly):
    seen = 0
    for name, body in kernels(assembly):
        loops = list(mfma_loops(body))
        assert loops, name
        for seg in loops:
            inasm, foreign = False, []
            for l in seg:
                t = l.strip()
                if t.startswith(";;#ASMSTART"): inasm = True; continue
                if t.startswith(";;#ASMEND"): inasm = False; continue
                if not t or t.startswith(";") or t.startswith("."): continue
                if not inasm and not ALLOWED.match(t.split(";")[0].strip()):
                    foreign.append(t.split(";")[0].strip())
            assert not foreign, "%s: the compiler emitted %s inside a k-step loop" % (name, Counter(x.split()[0] for x in foreign))
            seen += 1
    assert seen >= 13           # 4 + 2 (two heights in the SHORTS instantiations) + 3 round-3 kernels + 2 x 2 phases of the persistent kernel


def test_no_spills_no_scratch_and_the_register_budget(assembly):
    meta = re.findall(r"\.name:\s+(_Z\d+gram_i8p{0,2}_kernel\w+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", assembly, re.S)
    assert len(meta) >= 9
    for name, scratch, sspill, vgpr, vspill in meta:
        assert int(scratch) == 0 and int(vspill) == 0, (name, scratch, vspill)
        assert int(vgpr) <= 512, (name, vgpr)
        if "gram_i8p" in name:
            assert int(sspill) == 0, (name, sspill)


def test_persistent_gram_keeps_its_prefetch_registers_and_waits_to_itself(assembly):
    """gram_i8pp_kernel issues the NEXT tile's count fragments (asm `global_load_dwordx4`) in front of this tile's epilogue and waits for them at the
    head of the next tile.  Sound only while the compiler (a) touches none of those fragment registers other than to zero them before the first
    load, (b) emits no VMEM load of its own inside the epilogue -- a load of its own there makes it wait for vmcnt(0), i.e. for the prefetch, at
    the head of the epilogue: the tile's output slots are therefore loaded with the tile's prologue, and every compiler-emitted `s_waitcnt vmcnt`
    sits right behind those loads (or the exit's atomics) --, and (c) spills nothing.  (Block layout is the compiler's: the checks are per kernel.)"""
    found = 0
    for name, body in kernels(assembly):
        if "gram_i8pp" not in name:
            continue
        found += 1
        flags, inasm = [], False
        for l in body:
            t = l.strip()
            if t.startswith(";;#ASMSTART"): inasm = True
            flags.append(inasm)
            if t.startswith(";;#ASMEND"): inasm = False
        frag = set()
        for i, l in enumerate(body):
            mm = re.match(r"\s*global_load_dwordx4 v\[(\d+):(\d+)\]", l)
            if mm and flags[i]:
                frag.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
        assert len(frag) >= 32, name
        own_vmem = [i for i, l in enumerate(body) if not flags[i] and re.match(r"\s*(global|flat|buffer)_(load|atomic)", l)]
        stores = [i for i, l in enumerate(body) if not flags[i] and re.match(r"\s*(global|flat)_store", l)]
        assert stores and 4 <= len(own_vmem) <= 24, (name, len(own_vmem))       # output slots / scales of a tile (x 2 phases) + the exit's counters: nothing per store
        for i, l in enumerate(body):
            t = l.strip()
            if flags[i] or not t or t.startswith(";") or t.startswith("."): continue
            code = t.split(";")[0]
            assert not code.startswith("scratch_"), (name, code)
            if "s_waitcnt" in code and "vmcnt" in code:
                # the wait for its own loads (or the release in front of the exit's counter update), not one in the middle of an epilogue
                assert any(0 < i - k <= 48 for k in own_vmem) or i >= len(body) - 60, (name, i, code)
            touched = set()
            for mm in re.finditer(r"v\[(\d+):(\d+)\]", code): touched.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
            for mm in re.finditer(r"\bv(\d+)\b", code): touched.add(int(mm.group(1)))
            if touched & frag:
                # registers are shared between the phases (tall / short tiles) and with what precedes a phase: the only thing allowed on a register that
                # is a prefetch target is the zeroing in front of its first load and address arithmetic / epilogue temporaries of the OTHER phase;
                # what must never appear is a copy FROM such a register (a v_mov / v_accvgpr_write with it as the source)
                srcs = code.split(",", 1)[1] if "," in code else ""
                src_regs = set()
                for mm in re.finditer(r"v\[(\d+):(\d+)\]", srcs): src_regs.update(range(int(mm.group(1)), int(mm.group(2)) + 1))
                for mm in re.finditer(r"\bv(\d+)\b", srcs): src_regs.add(int(mm.group(1)))
                dst_regs = touched - src_regs
                if code.startswith("v_mov") or code.startswith("v_accvgpr_write"):
                    # (zeroing chains in front of a phase copy one fragment register into another: allowed; a fragment's content must never LEAVE the set)
                    assert not (src_regs & frag) or (dst_regs and dst_regs <= frag), (name, code)
    assert found == 2


def test_filler_schedule_tables():
    """The constexpr schedule of gram_i8p_kernel (GramI8PStep): every filler behind an MFMA of its own k-step, a count load never before the MFMAs that
    read its register have been issued, the fragment reads back long before the barrier -- for every instantiation the library builds."""
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc in this environment")
    prog = r'''
#include <hip/hip_runtime.h>
#include <cstdio>
typedef int i32x4 __attribute__((ext_vector_type(4)));
__device__ __forceinline__ void glds_block(const void*, unsigned, unsigned) {}
#include "%(csrc)s/kernels_gram_i8p.h"
template <int S, int MW> void show() {
    using K = GramI8PStep<S, MW, 2>;
    printf("%%d %%d %%d %%d |", S, MW, K::NMFMA, K::PERB);
    for (int i = 0; i < MW; ++i) printf(" %%d", K::aslot(i));
    printf(" |"); for (int i = 0; i < K::PERB; ++i) printf(" %%d", K::dslot(i));
    printf(" |"); for (int i = 0; i < K::NB; ++i) printf(" %%d", K::rslot(i));
    printf("\n");
}
int main() { show<6, 5>(); show<6, 4>(); show<7, 4>(); show<7, 3>(); return 0; }
''' % {"csrc": CSRC}
    with tempfile.TemporaryDirectory() as d:
        src = os.path.join(d, "sched.hip")
        open(src, "w").write(prog)
        exe = os.path.join(d, "sched")
        subprocess.check_call([HIPCC, "-O1", "-std=c++17", "-Wno-unused-value", "-o", exe, src], stderr=subprocess.DEVNULL)
        lines = subprocess.check_output([exe], text=True).strip().split("\n")
    assert len(lines) == 4
    for line in lines:
        head, a, d_, r = [x.split() for x in line.split("|")]
        S, MW, nmfma, perb = map(int, head)
        a, d_, r = list(map(int, a)), list(map(int, d_)), list(map(int, r))
        NB = 2 * S
        assert len(a) == MW and len(d_) == perb and len(r) == NB
        slots = a + d_ + r
        assert len(set(slots)) == len(slots) and all(0 < x < nmfma for x in slots), line       # one filler per MFMA gap
        assert all(a[i] >= NB * i + 1 for i in range(1, MW)), line                               # reload behind the tile that read the register
        assert max(r) <= nmfma - 10, line                                                        # >= 160 clocks of MFMAs behind the last fragment read (an LDS round trip is ~130)
        vm = sorted(a + d_)
        assert max(y - x for x, y in zip(vm, vm[1:])) <= 2 * nmfma // len(vm) + 2, line          # VMEM evenly spread


# ------------------------------------------------------------------------------------------------ the one-wave / four-wave solver kernels (round 5)
# These kernels hold a 64-column window of the covariance in 128 registers and are latency-bound chains of small phases: a value the allocator sends
# to scratch inside the treatment loop or the iteration costs a memory round trip per use (the first quad kernel reloaded the column sum, 1 / n and the
# scale factor three times per column: 59 k of a problem's 169 k clocks).  The test compiles the solver unit as the release build does and holds the
# spill counts of the register-resident kernels at what the committed source gives (+ a margin for compiler noise).
SOLVER_LIMITS = {  # kernel substring -> (max spilled VGPRs, max scratch bytes per lane)
    "solver_wave16_kernelILi8ELb0E": (8, 48),
    "solver_wave16_kernelILi8ELb1E": (12, 64),
    "solver_wave16_kernelILi16ELb0E": (28, 128),
    "solver_wave16_kernelILi16ELb1E": (28, 128),
    "solver_quad_kernelILi16E": (10, 64),
}
SOLVER_ONE_WAVE = {"solver_wave16_kernelILi32ELb0E": (0, 0)}      # one wave per SIMD (three problems per CU by LDS): 512 registers, nothing in scratch


def test_solver_kernels_keep_their_windows_in_registers():
    if not os.path.exists(HIPCC):
        pytest.skip("no hipcc in this environment")
    with tempfile.TemporaryDirectory() as d:
        r = subprocess.run([HIPCC, "--offload-arch=gfx950", "-O3", "-std=c++17", "-fPIC", "-Wno-unused-value", "-Wno-unused-result", "--cuda-device-only", "-c",
                            "-Rpass-analysis=kernel-resource-usage", os.path.join(CSRC, "plspm_fit.hip"), "-o", os.path.join(d, "fit.o")], capture_output=True, text=True)
        assert r.returncode == 0, r.stderr[-2000:]
    usage, name = {}, None
    for line in r.stderr.splitlines():
        m = re.search(r"Function Name: (\S+)", line)
        if m:
            name = m.group(1); usage[name] = {}
        for key in ("VGPRs Spill", "ScratchSize [bytes/lane]", "Occupancy [waves/SIMD]", "VGPRs"):
            m = re.search(re.escape(key) + r": (\d+)", line)
            if m and name:
                usage[name].setdefault(key, int(m.group(1)))
    for sub, (max_spill, max_scratch) in SOLVER_LIMITS.items():
        found = [k for k in usage if sub in k]
        assert len(found) == 1, (sub, found)
        u = usage[found[0]]
        assert u["Occupancy [waves/SIMD]"] == 2, (sub, u)                   # eight one-wave problems / two four-wave problems per CU
        assert u["VGPRs Spill"] <= max_spill and u["ScratchSize [bytes/lane]"] <= max_scratch, (sub, u)
    for sub, (max_spill, max_scratch) in SOLVER_ONE_WAVE.items():
        found = [k for k in usage if sub in k]
        assert len(found) == 1, (sub, found)
        u = usage[found[0]]
        assert u["Occupancy [waves/SIMD]"] == 1 and u["VGPRs Spill"] <= max_spill and u["ScratchSize [bytes/lane]"] <= max_scratch, (sub, u)
