// chol_kernels.h -- launchers of the tile Cholesky / triangular-solve / PCG kernels.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "ba_kernels.h"

namespace apex {

struct GemmTask {  // C = beta*C + alpha * A * B^T on 144x144 row-major tiles
    double* C;
    const double* A;
    const double* B;
};

struct PotrfTask {  // one diagonal tile: factor in place, inverse of the factor to Linv
    double* A;
    double* Linv;
    int K;
};

constexpr int kFlowFirstWriter = 16;   // FactorUnit::kind bit: the unit is the FIRST writer of a fill tile (nothing is read from the target)
constexpr int kFlowUnitsPerTile = 9;   // units per panel solve / update; the weight of a potrf in the version counters
struct FactorUnit {   // one workgroup of the dataflow factorisation of the top of the elimination tree (k_factor_flow)
    double* C;            // potrf: the diagonal tile (factorised in place); product: the target tile
    const double* A;      // potrf: L^-1 of the tile (written); product: left operand tile (panel solve: == C, in place)
    const double* B;      // product: right operand tile (panel solve: L^-1 of the column's diagonal tile)
    int wait_flag[3];     // indices into the version array, -1 = none: [0] the target's previous writer, [1] A final, [2] B final
    int wait_val[3];      // ... proceed when ver[flag] >= val
    int pub;              // index into the version array: += 1 per finished unit (kFlowUnitsPerTile per tile and writer), += 9 by a potrf
    int kind;             // 0 potrf + inverse, 1 panel solve C = A B^T (in place), 2 update C -= A B^T (one 48 x 48 block), 3 update, the whole tile (publishes 9)
    int strip;            // panel solve: 16-row strip 0..8 of C; update: 48 x 48 block 3 bi + bj of C; potrf: tile column (for the failure flag)
    int pad;
};

struct TriTask {   // one workgroup of a triangular-solve step (k_tri_step)
    const double* Mdiag;  // Linv of the step's diagonal tile
    const double* Moff;   // the off-diagonal tile this workgroup applies (unused when other < 0)
    int k;                // block solved in this step
    int other;            // block updated by this workgroup; -1: store the solved block instead
};

struct FlowTask {     // one workgroup of a dataflow triangular sweep (k_tri_fwd_flow / k_tri_bwd_flow)
    const double* mat;    // product task: the off-diagonal tile; solve task: L^-1 of the diagonal tile
    int src;              // product: the block whose solution the tile multiplies; solve: -1; fold only (distributed
                          // forward, a shared top block: right-hand side minus this rank's products, no solve): -2
    int dst;              // product: the block the product belongs to; solve: the block solved
    int part;             // product: its slot in the partial array; solve: first slot of the block's products
    int count;            // solve: number of products to wait for and fold
    // solve task, round 5 (single-GPU plans): the product of the block's LAST-ARRIVING source -- the link of the dependency chain --
    // is formed by the solve task itself (tile mat2 times the solution of block src2, into slot `slot2` of the fold, same
    // arithmetic, same place in the sum): one flag hop and one trip of the product through memory less per level.  src2 < 0: none
    const double* mat2 = nullptr;
    int src2 = -1;
    int slot2 = -1;
};

struct SymEntry {  // one tile of block-row I of the symmetric tile matrix
    int slot;      // tile slot
    int other;     // the other block index (column block for kind 0/2, row block for kind 1)
    int kind;      // 0: tile (I,other) other<I ; 1: tile (other,I) other>I (use transpose) ; 2: diagonal
};

struct SymTile { int slot, I, J; };  // a structurally non-zero tile (I >= J) of S

void launch_sym_tile_products(const SymTile* list, int n, const double* tiles, const double* x, double* part, hipStream_t s);
void launch_sym_tile_gather(int nt, const int* row_ptr, const SymEntry* entries, const double* part, const double* p,
                            double* y, double* row_dot, hipStream_t s);
void launch_pcg_step1(int n, int nt, const double* scal, const double* row_dot, const double* p, const double* ap,
                      const double* pre, double* x, double* r, double* blk_part, double* out_pap, hipStream_t s);
void launch_pcg_step2(int n, double* scal, const double* blk_part, const double* pre, const double* r, double* p,
                      double* out2, double abs_tol, hipStream_t s);
void launch_potrf_inv(const PotrfTask* tasks, int n, int* fail, hipStream_t s, int* arrived = nullptr);
void launch_gate(const int* arrived, int expected, int max_micros, hipStream_t s);
// the dataflow factorisation: one workgroup per unit, dispatched in list order; ver[] must be zero; err: error word (time-out)
void launch_factor_flow(const FactorUnit* units, int n_units, int* ver, int* fail, int* err, hipStream_t s,
                        unsigned long long* trace = nullptr);
// batches of <= 56 tasks use the latency kernels, larger ones the four-wave strip kernel (three workgroups per tile)
// tri_b: every B is a lower-triangular inverse written by launch_potrf_inv (zero 16 x 16 blocks right of the diagonal): the
// large-batch kernel then skips the 36 of 81 block products that multiply by them.
void launch_tile_gemm_nt(const GemmTask* tasks, int n, double alpha, double beta, hipStream_t s, bool tri_b = false);
// poison_block >= 0 (tests only): that block's counter is made unreachable after the flags are cleared, so the task that
// waits for it runs into the spin limit -- the time-out path (error word raised, wrong result) on demand
void launch_clear_i32(int* p, int64_t n, hipStream_t s);                   // a small clear as ONE kernel (chol_kernels.hip)
void launch_post_word(int* word, int* host_word_dev, hipStream_t s);       // *host = *word, *word = 0 when *word != 0
void launch_tri_flow(bool backward, const FlowTask* tasks, int n_tasks, const double* in, double* out, double* part, int* flags,
                     int nt, hipStream_t s, const double* fold_b, double* fold_out, int poison_block = -1, bool keep_flags = false);   // keep_flags: the second part of a sweep launched in two (the counters of the first part stand)
// tests only: n workgroups that each take a whole CU's LDS (nothing else that needs LDS fits beside them) and spin for
// `micros`; *started (host-visible) counts the workgroups that are resident
void launch_occupy_cus(int n, int micros, int* started, hipStream_t s);
void launch_tri_step(bool trans, const TriTask* tasks, int n, double* vwork, double* vout, hipStream_t s);
void launch_tile_diag(const double* tiles, const int* diag_slot, int nt, double* diag, hipStream_t s);
void launch_tile_add_diag(double* tiles, const int* diag_slot, int n_valid, int n_total, double add_valid,
                          double set_pad, hipStream_t s);
// distributed triangular solves: per-tile class masks (bit 1 << cls[tile]); in must not alias out for select
void launch_vec_select(int n, const double* in, const int* cls, int mask, double* out, hipStream_t s);
void launch_vec_merge(int n, const double* src, const int* cls, int mask, double* dst, hipStream_t s);
void launch_tile_scale_sym(const SymTile* list, int n, double* tiles, const double* scale /* n_pad */, hipStream_t s);
void launch_pcg_init(int n, const double* diag, const double* b, double* pre, double* x, double* r, double* z, double* p,
                     hipStream_t s);
void launch_dot(int n, const double* a, const double* b, double* out, hipStream_t s);
void launch_pcg_update_xr(int n, double alpha, const double* p, const double* ap, double* x, double* r, hipStream_t s);
void launch_pcg_update_xr_dev(int n, double rz_old, const double* pap /* device */, const double* p, const double* ap, double* x, double* r, hipStream_t s);
void launch_pcg_update_p(int n, double beta, const double* z, double* p, hipStream_t s);
// the matrix-free PCG's scalars on the device (sc: [0] r.r [1] r.z [2] p.Ap [4] rz_old [5] frozen [6] beta): see k_pcg_implicit_close
void launch_pcg_implicit_begin(double* sc, hipStream_t s);
void launch_pcg_update_xr_sc(int n, const double* sc, const double* p, const double* ap, double* x, double* r, hipStream_t s);
void launch_pcg_implicit_close(double* sc, double abs_tol, hipStream_t s);
void launch_pcg_update_p_sc(int n, const double* sc, const double* z, double* p, hipStream_t s);

}  // namespace apex
