#!/bin/bash
# scripts/isa_census.sh: profiles/r06_isa_census_*.txt of the two kernels the headline launches, from the sources as they are
# (hipcc -S cross-compiles without a GPU; seconds): the slim fill kernel alone in a translation unit of its own, the relaxation
# instances from tests/asm/relax_instances.hip.  MEASUREMENT TOOLING (tools/isa_stats.py, tools/isa_loops.py), not a product path.
set -e
ROOT=$(cd "$(dirname "$0")/.." && pwd)
W=$(mktemp -d /tmp/kascensus.XXXXXX)
SHA=$(cd "$ROOT" && python -c "import bench; print(bench.sources_sha16())")
HIPCC="/opt/rocm/bin/hipcc --offload-arch=gfx950 -O3 -std=c++17 --cuda-device-only -S -I$ROOT/include -I$ROOT/kafka-assigner_amd/csrc"
cat > $W/slim.hip <<'SRC'
#define KAS_ABI_FN __host__ __device__ static inline
#include <hip/hip_runtime.h>
#include "kas_abi.h"
#include "kas_plan_math.h"
#include "kas_solver_body.h"
template <int W, bool M32>
__global__ __launch_bounds__(256, 4) void kas_fill_slim_kernel(KasLaunch a) {
  extern __shared__ __attribute__((aligned(16))) unsigned char kas_lds[];
  for (int32_t s = (int32_t)blockIdx.x; s < a.n_scenarios; s += (int32_t)gridDim.x)
    kas::fill_scenario<W, 4, true, M32 ? 1 : 0>(a, s, kas_lds);
}
template __global__ void kas_fill_slim_kernel<3, true>(KasLaunch);
SRC
$HIPCC -o $W/slim.s $W/slim.hip 2>/dev/null
$HIPCC -o $W/relax.s $ROOT/tests/asm/relax_instances.hip 2>/dev/null
res() { grep -E "^; (NumVgprs|ScratchSize|Occupancy):" $1 | head -3 | tr '\n' ' ' | sed 's/; //g'; }
{
echo "# ISA census of kas_fill_slim_kernel<3, M32=true> — the headline's fill kernel since round 6, the instance for dword mid rows (int32 cells, per-chunk histograms, a direct id table, the quota drawn with the"
echo "# atomic-with-return, first fit handed over): gfx950, hipcc -O3; kernel sources sha16 $SHA.  $(res $W/slim.s)"
echo "# (kas_fill_kernel<3,4>, which holds every path: 5,269 basic blocks / 69,034 instructions, 128 VGPRs + 384 B of scratch per lane)."
echo "# tools/isa_stats.py totals, then tools/isa_loops.py: every basic block inside a loop of depth >= 2 (scenario loop > topic loop > row-tile loop) with its memory signature."
echo "# How to read it: the first row scan (pass A) is the header block with 4 x global_load_dwordx3 (four tiles of rows asked for per lane) and the blocks behind it with"
echo "# ds_read_u16 (id -> index) / ds_read_i16 (rack) / ds_add_u32 (per-chunk histogram); the second scan (pass B) the blocks with ds_add_rtn_u32 (quota draw) and"
echo "# global_store_dword (the dword mid row: v_med3_u32 / v_min3_u32 in front of it sort the holders); the blocks with ds_write_b16 / ds_write2_b32 in front are the node tables going into the LDS."
python $ROOT/tools/isa_stats.py $W/slim.s kas_fill_slim | head -1
python $ROOT/tools/isa_loops.py $W/slim.s kas_fill_slim 2
} > $ROOT/profiles/r06_isa_census_fill_slim_kernel_3.txt
K=_Z22kas_order_relax_kernelILi3ELb0ELb0ELb0ELb0ELb1ELb1ELb0EEv9KasLaunch
{
echo "# ISA census of kas_order_relax_kernel<3, DUAL=false, CTX=false, VERIFY=false, C16=false, IDL=true, M32=true> - the headline's order kernel (int32 cells, ids in the LDS, tiles of 64 rows, dword mid rows);"
echo "# gfx950, hipcc -O3; kernel sources sha16 $SHA.  tools/isa_stats.py --blocks, then tools/isa_loops.py (depth >= 2: topic loop > tile loop > evaluation loop)."
echo "# The fast path (64 rows of three holders) = the block with 1 x global_load_dword in front (the row's dword), the header with 3 x ds_add_rtn_u32 and v_min3_u32 (relax_eval3 on constant tags), and the depth-3 loop with 3 x ds_sub_u32 + 3 x ds_add_rtn_u32 (relax_pairs<3>);"
echo "# the slow path (rows with fewer than three holders) is the pair with v_min_u32 clamps and the rank computation."
python $ROOT/tools/isa_stats.py $W/relax.s $K --blocks
python $ROOT/tools/isa_loops.py $W/relax.s $K 2
} > $ROOT/profiles/r06_isa_census_order_relax_3_idl.txt
rm -rf $W
echo "census for sources $SHA"
