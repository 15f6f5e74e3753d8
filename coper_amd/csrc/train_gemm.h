// Split-16 GEMMs of the training step (train_gemm_bf16.hip; the file names are round 2's): operand packing and
// C(i,j) = sum_k X(i,k) Y(j,k).  Round 5: the split is the inference path's SCALED fp16 split (split16.h) -- every packed operand
// carries an exact power of two chosen from its own largest magnitude (gradients are 1e-6, activations 1e+1: nothing puts them in
// fp16's window), 22 bits per value instead of the 16 of round 2's bf16 split: 5e-7 relative per product instead of 1.5e-5.
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

struct coper_handle;

namespace coper {

// two-level strided index: offset(i) = (i / seg) * s_hi + (i % seg) * s_lo   (seg == 0: i * s_lo)
struct TgIdx {
  int64_t seg, s_hi, s_lo;
};
inline TgIdx tg_idx(int64_t stride) { return TgIdx{0, 0, stride}; }
inline TgIdx tg_idx2(int64_t seg, int64_t s_hi, int64_t s_lo) { return TgIdx{seg, s_hi, s_lo}; }

// hi / lo bf16 planes in the fragment order of v_mfma_f32_32x32x16_bf16: [rows_pad / 32][tg_ks_stride(K)][64 lanes] x 16 B each
struct TgPlanes {
  uint4* hi = nullptr;
  uint4* lo = nullptr;
  int32_t* exp = nullptr;   // device word: the planes hold X 2^exp (written by tg_pack, read by the GEMM's epilogue)
};

constexpr int TG_SUMSQ_SLOTS = 64;
constexpr int64_t TG_ROW_PAD = 128;   // rows of a plane set are padded to the GEMM's workgroup tile
inline int64_t tg_rows_pad(int64_t rows) { return (rows + TG_ROW_PAD - 1) / TG_ROW_PAD * TG_ROW_PAD; }
// k-steps between two row blocks of a plane: odd, so that the fragments a wave reads in one k-step (row blocks a, a+1, ...: same
// k-step, one block stride apart) do not all sit on the same few memory channels -- with strides of 288 / 32 / 400 KiB (the
// FB15k-237 shapes) every wave of the chip hit the same quarter of the channels in every k-step
inline int64_t tg_ks_stride(int64_t K) { return ((K + 15) / 16) | 1; }
inline size_t tg_plane_elems(int64_t rows, int64_t K) { return (size_t)(tg_rows_pad(rows) / 32) * (size_t)tg_ks_stride(K) * 64; }   // uint4 per plane

// X(row, k) = src[off(ri, row) + off(ki, k)] -> planes (rows zero-padded to R_pad, k to a multiple of 16).
// rows_fast: consecutive rows are contiguous in memory (the pack reads along rows), else consecutive k are.
// exp_from: another plane set packed from the SAME tensor earlier (its exponent is reused: no second pass for the maximum);
// otherwise the largest |X| is reduced first (scratch: two device words)
// max_slots: TG_MAX_SLOTS device words whose maximum is the bit pattern of the operand's largest |X| -- left by the kernel that
// PRODUCED the tensor (the optimizer's pass over the dense weights: coper_train.hip) -- the pack reduces them itself: no pass
// over the tensor for its maximum at all
constexpr int TG_MAX_SLOTS = 1024;
int tg_pack(coper_handle* h, const float* src, TgIdx ri, TgIdx ki, int64_t R, int64_t K, int64_t R_pad, bool rows_fast, TgPlanes out,
            hipStream_t s, unsigned* scratch, const int32_t* exp_from = nullptr, const unsigned* max_slots = nullptr);
// Both views of one tensor from ONE read: view A = rows i (ri) contracted over k (ki) into `a`; view B = rows k contracted over i into
// `b` (the operand of the product that contracts the other index).  ki must be 16-byte loadable (unit stride, multiples of four).
int tg_pack_both(coper_handle* h, const float* src, TgIdx ri, TgIdx ki, int64_t R, int64_t K, TgPlanes a, TgPlanes b, hipStream_t s,
                 unsigned* scratch, const unsigned* max_slots = nullptr);
// C[off(ci, i) + off(cj, j)] = sum_k X(i, k) Y(j, k), i < M, j < N
// nsplit > 1: K is cut into nsplit slices whose partial sums go to `part` ([nsplit][M][N] floats) and are summed in slice order
// sumsq: when not null, the sum of the squares of the stored C is added by the storing kernel to the TG_SUMSQ_SLOTS device
// doubles at sumsq (a workgroup adds to slot (its index) % TG_SUMSQ_SLOTS: thousands of atomics on one address serialise)
int tg_gemm_nt(coper_handle* h, TgPlanes X, int64_t M, TgPlanes Y, int64_t N, int64_t K, float* C, TgIdx ci, TgIdx cj, hipStream_t s,
               int nsplit = 1, float* part = nullptr, double* sumsq = nullptr, bool leave_slices = false);
// (leave_slices: with nsplit > 1 the partial sums stay in `part` and C is not written -- the caller's next kernel adds them, in slice order)
// slices that fill the chip when the output has few 128 x 128 tiles and K is long (1: no split)
int tg_split_k(int64_t M, int64_t N, int64_t K);

}  // namespace coper
