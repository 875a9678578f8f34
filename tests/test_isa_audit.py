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
    for m in re.finditer(r"^(_Z\d+gram_i8p?_kernel\w+):[^\n]*\n(.*?)s_endpgm", asm, re.S | re.M):
        yield m.group(1), m.group(2).split("\n")


def mfma_loops(body):
    labels = {l.split(":")[0].strip(): i for i, l in enumerate(body) if re.match(r"\.LBB\w+:", l.strip())}
    for i, l in enumerate(body):
        mm = re.match(r"\s*s_c?branch\w*\s+(\.LBB\w+)", l)
        if mm and mm.group(1) in labels and labels[mm.group(1)] < i:
            seg = body[labels[mm.group(1)]:i + 1]
            if any("v_mfma" in x for x in seg):
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
    assert seen >= 9            # 4 + 2 (two heights in the SHORTS instantiations) + 3 round-3 kernels


def test_no_spills_no_scratch_and_the_register_budget(assembly):
    meta = re.findall(r"\.name:\s+(_Z\d+gram_i8p?_kernel\w+).*?\.private_segment_fixed_size:\s+(\d+).*?\.sgpr_spill_count:\s+(\d+).*?\.vgpr_count:\s+(\d+).*?\.vgpr_spill_count:\s+(\d+)", assembly, re.S)
    assert len(meta) >= 7
    for name, scratch, sspill, vgpr, vspill in meta:
        assert int(scratch) == 0 and int(vspill) == 0, (name, scratch, vspill)
        assert int(vgpr) <= 512, (name, vgpr)
        if "gram_i8p" in name:
            assert int(sspill) == 0, (name, sspill)


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
