// sw_batch_kernels.h — device-side job lists for batches of MANY SMALL whole problems (the UniProt shape of
// src/mpi_sw_solve_uniprot.cpp:95-138: half a million database sequences, each against one short query).
//
// The host used to build one descriptor per alignment per pass (problem lists up, results down, walk descriptors up,
// lengths down, offsets up: ~250 MB of staging and several per-item host loops for 561 k alignments).  Here the
// descriptors of sw_wave_kernel / sw_wave_walk_kernel are written ON THE DEVICE from what is already resident — the
// batch's offsets, lengths and length-sorted ids — and the only things that cross PCIe are the results:
//   batch_wave_setup   WaveProblem[k] for sorted positions [first, first + count): whole problem, TRACK + DIRS
//   (sw_wave_kernel)   one pass: first maximum in storage order AND the greedy decision of every cell
//     or, lanes = columns of the shared second sequence with dyadic scores (sw_wave_prof_kernel, host_batch.h):
//     (TRACK pass)       first maximum, the slot's wavefront saved every kCkptEvery steps
//     batch_window_setup WaveProblem[k] <- the rows [k0, row of the argmax), resumed from the state saved at step k0
//     (DIRS pass)        decisions for those rows only (a walk that leaves them: the problem goes to the host-driven path)
//   batch_walk_setup   WaveWalk[k] from the argmax the pass just found
//   (sw_wave_walk_kernel<kWalkMeasure>)  ->  walk_sizes  ->  exclusive scan  ->  (sw_wave_walk_kernel<kWalkWrite>)
// plus a three-kernel exclusive scan of int64 (block sums, scan of the sums, add).
#pragma once
#include <hip/hip_runtime.h>
#include <stdint.h>

#include "sw_wave_kernel.h"

namespace mi355sw {

struct BatchWaveArgs {
  const uint8_t *qbytes;     // resident queries
  const int64_t *qoff;
  const int32_t *qlen;
  const int32_t *qsel;       // query ids sorted by length
  const int64_t *qcum;       // [nq + 1] exclusive prefix of the lengths in sorted order
  const uint8_t *ref;        // first byte of the range of the reference
  int64_t nref;
  int first, count;          // sorted positions [first, first + count)
  int orient;                // as WaveProblem: 0 lanes = rows of x (stream = the range), 1 lanes = columns of y (stream = x)
  int W;                     // dwords of decisions per lane and stream position
  uint32_t *dirs;            // decision scratch of this launch
  WaveProblem *probs;        // [count]
  float *best;               // [count]
  int64_t *cell;             // [2 * count]
  // checkpointed whole problems (orient 1 on sw_wave_prof_kernel, host_batch.h): the first pass keeps (best, cell) and saves
  // the wavefront's state every kCkptEvery steps; batch_window_setup then turns every problem into its last kWindowGuard + 1 .. + kCkptEvery rows in
  // front of the argmax, resumed from the saved state, and only those rows get decisions
  float *ckpt;               // null: whole problems in one pass
  int f16;                   // the first pass ran on sw_wave_prof16_kernel (two problems per slot, one set of saved states per pair)
  int R;
  // Long streams cut into PIECES (orient 1 on sw_wave_prof_kernel, host_batch.h): a launch is at least as long as its longest
  // stream, and a 7 k-residue sequence is 20 average ones — a fixed cost that does not shrink with a rank's share of the
  // database.  The first `npieces` problems of the launch are pieces of the `nlong` longest sequences (launch order = longest
  // first): piece e = rows [pc_start[e], pc_start[e] + pc_rows[e]) of sequence pc_seq[e], pc_before[e] stream positions in
  // front of it; the other count - nlong problems are whole sequences.  A piece starts kPieceWarm-ish rows in front of its own
  // rows (the margin of DESIGN.md L1 along the stream + the decision window's reach), so its own rows — and every checkpoint
  // a decision window in front of an own row resumes from — are exact; results are merged per sequence by batch_seq_results.
  int nlong, npieces, piece_rows;
  const int32_t *pc_seq, *pc_start, *pc_rows, *pc_first;   // pc_first[k], k = 0 .. nlong: first piece of sequence k
  const int64_t *pc_before;
  int64_t piece_stream, long_stream;                       // stream positions of all pieces / of the nlong sequences as wholes
  float *sbest;              // per SEQUENCE (launch order), after batch_seq_results: maximum,
  int64_t *scell;            // ... its first cell,
  int32_t *sprob;            // ... and the problem (piece) whose own rows hold that cell: the one a decision window resumes in
  WaveProblem *probs2;       // [count] the decision windows (batch_window_setup)
};

constexpr int kPieceRows = 1024;                                   // own rows of a piece of a long stream (BatchWaveArgs)
// rows in front of the argmax a window holds at least.  Config 4's walks (unrelated sequences): median 4 rows, 18 at the 99.9th
// percentile, 23 the longest of 20 000; a walk that leaves its window takes the host-driven path (left_window)
constexpr int kWindowGuard = 32;
constexpr int kWindowRows = kWindowGuard + kCkptEvery + 16;        // decision rows per window: guard + 1 .. guard + kCkptEvery rows, + the skew

// saved states in front of problem k: one row of 16 x (R + 1) floats per kCkptEvery steps (as batch_dirs_offset: monotone in k,
// room for (steps_k + 16) / kCkptEvery + 1 rows)
__device__ __host__ inline int64_t batch_ckpt_row(int64_t stream_positions_before, int64_t k) {
  return (stream_positions_before + 16 * k) / kCkptEvery + k;
}

// decision bytes in front of problem k of the launch (k = 0 .. count): (stream positions so far + 16 rows of skew per
// problem) * 16 lanes * W dwords (as wave_dirs_bytes on the host)
__device__ __host__ inline int64_t batch_dirs_offset(int64_t stream_positions_before, int64_t k, int W) {
  return (stream_positions_before + 16 * k) * 16 * (int64_t)W * 4;
}

// sorted position of problem k of a launch over sorted positions [first, first + count): longest first (see batch_wave_setup)
__device__ __host__ inline int batch_sorted_pos(int first, int count, int k) { return first + count - 1 - k; }

__global__ void batch_wave_setup(const BatchWaveArgs a) {
  const int kk = blockIdx.x * blockDim.x + threadIdx.x;            // problem of the launch: a piece, or a whole sequence
  if (kk >= a.npieces + a.count - a.nlong) return;
  const bool piece = kk < a.npieces;
  const int k = piece ? a.pc_seq[kk] : a.nlong + (kk - a.npieces); // its sequence (launch order)
  // LONGEST FIRST: problem k of the launch is sorted position first + count - 1 - k.  Workgroups are dispatched in index order,
  // and a launch is at least as long as its longest stream (a 7 k-residue sequence: 0.8 ms on one 16-lane slot); started
  // last it ran alone at the end of the launch — a fixed cost that does not shrink with a rank's share of the database
  const int sp = batch_sorted_pos(a.first, a.count, k);
  const int id = a.qsel[sp];
  const uint8_t *xq = a.qbytes + a.qoff[id];
  const int32_t m = a.qlen[id];
  WaveProblem w;
  int64_t before;                                                  // stream positions of the problems in front of this one
  w.b_offset = 0;
  if (a.orient == 0) { w.a = xq; w.na = m; w.b = a.ref; w.nb = (int32_t)a.nref; before = (int64_t)k * a.nref; }
  else if (piece) { w.a = a.ref; w.na = (int32_t)a.nref; w.b = xq + a.pc_start[kk]; w.nb = a.pc_rows[kk]; w.b_offset = a.pc_start[kk]; before = a.pc_before[kk]; }
  else { w.a = a.ref; w.na = (int32_t)a.nref; w.b = xq; w.nb = m; before = a.piece_stream + (a.qcum[a.first + a.count] - a.qcum[sp + 1]) - a.long_stream; }
  w.dirs = a.ckpt != nullptr ? nullptr
                             : reinterpret_cast<uint32_t *>(reinterpret_cast<uint8_t *>(a.dirs) + batch_dirs_offset(before, kk, a.W));
  w.ckpt = a.ckpt != nullptr ? a.ckpt + (size_t)batch_ckpt_row(before, kk) * 16 * (size_t)(a.R + 1) : nullptr;
  w.k0 = 0; w.ck_half = 0; w.lanes_used = 0;
  w.best = a.best + kk;
  w.cell = a.cell + 2 * (size_t)kk;
  w.target = 0.0f; w.own_lo = 0; w.full_n = a.nref;
  a.probs[kk] = w;
}

// per sequence k (launch order): the maximum over its pieces, the first cell holding it in the float engine's storage order
// (column of y, then row of x; a cell seen by two overlapping pieces is the same cell), and the piece whose OWN rows hold that
// cell; whole sequences copy their problem's result
__global__ void batch_seq_results(const BatchWaveArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.count) return;
  if (k >= a.nlong) {
    const int kk = a.npieces + (k - a.nlong);
    a.sbest[k] = a.best[kk];
    a.scell[2 * (size_t)k] = a.cell[2 * (size_t)kk];
    a.scell[2 * (size_t)k + 1] = a.cell[2 * (size_t)kk + 1];
    a.sprob[k] = kk;
    return;
  }
  float bv = 0.0f;
  int64_t bi = 0, bj = 0;
  const int e0 = a.pc_first[k], e1 = a.pc_first[k + 1];
  bool undecided = false;                                          // a piece the float16 pass left undecided (best = -1): so is the sequence
  for (int e = e0; e < e1; ++e) {
    const float v = a.best[e];
    undecided |= v < 0.0f;
    const int64_t i = a.cell[2 * (size_t)e], j = a.cell[2 * (size_t)e + 1];
    if (v > bv || (v == bv && v > 0.0f && (j < bj || (j == bj && i < bi)))) { bv = v; bi = i; bj = j; }
  }
  if (undecided) bv = -1.0f;
  a.sbest[k] = bv;
  a.scell[2 * (size_t)k] = bv > 0.0f ? bi : 0;
  a.scell[2 * (size_t)k + 1] = bv > 0.0f ? bj : 0;
  int own = bv > 0.0f ? (int)((bi - 1) / a.piece_rows) : 0;
  if (own > e1 - e0 - 1) own = e1 - e0 - 1;
  a.sprob[k] = e0 + own;
}

// after the first pass of a checkpointed launch: problem k becomes the rows [k0, row of its argmax) with decisions, resumed from
// the state saved at step k0 (whole from step 0 when the argmax lies within the first window; nothing when no cell is positive)
__global__ void batch_window_setup(const BatchWaveArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.count) return;
  WaveProblem w = a.probs[a.sprob[k]];                             // (a piece: rows count from its first row, b_offset)
  const bool hit = a.sbest[k] > 0.0f;
  const int32_t rows = hit ? (int32_t)(a.scell[2 * (size_t)k] - w.b_offset) : 0;   // 1-based row of the argmax = rows to run
  const int32_t k0 = hit ? kCkptEvery * (max(0, rows - 1 - kWindowGuard) / kCkptEvery) : 0;
  w.nb = rows;
  w.k0 = k0;
  // (states of the packed float16 pass: one set per PAIR of problems, at the first one's place)
  float *states = a.f16 ? a.probs[a.sprob[k] & ~1].ckpt : w.ckpt;
  w.ckpt = k0 > 0 ? states + (size_t)(k0 / kCkptEvery - 1) * 16 * (size_t)(a.R + 1) : nullptr;
  w.ck_half = a.f16 ? 1 + (a.sprob[k] & 1) : 0;
  // (lanes hold columns of y: the walk starts in the argmax column's lane and never moves right)
  w.lanes_used = hit ? (int32_t)((a.scell[2 * (size_t)k + 1] - 1) / a.R) + 1 : 1;
  w.dirs = a.dirs + (size_t)k * kWindowRows * 16 * (size_t)a.W;
  a.probs2[k] = w;
}

struct BatchWalkArgs {
  const WaveProblem *probs;  // [count] as built by batch_wave_setup
  const uint8_t *qbytes;
  const int64_t *qoff;
  const int32_t *qsel;
  const uint8_t *ref;
  int first, count, orient, R;
  const float *best;
  const int64_t *cell;
  WaveWalk *walks;           // [count]
  int64_t *wout;             // [3 * count]
};

__global__ void batch_walk_setup(const BatchWalkArgs a) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.count) return;
  const WaveProblem P = a.probs[k];
  const int id = a.qsel[batch_sorted_pos(a.first, a.count, k)];
  WaveWalk w;
  w.x = a.qbytes + a.qoff[id];
  w.y = a.ref;
  w.dirs = P.dirs;
  w.na = P.na; w.nb = P.nb; w.orient = a.orient;
  w.R = a.R; w.lanes = 16; w.skew = 1; w.row0 = P.k0;
  w.need_slope = 0.0f;
  w.b_offset = P.b_offset;                                         // (a piece of a long stream: its first row)
  const bool hit = a.best[k] > 0.0f;
  w.start_i = hit ? a.cell[2 * (size_t)k] : 0;                     // no positive cell: the walk emits nothing
  w.start_j = hit ? a.cell[2 * (size_t)k + 1] : 0;
  w.exact_from = 0;                                                // whole problem: every cell is exact
  w.cap = P.na + P.nb + 2;
  w.out = a.wout + 3 * (size_t)k;
  w.zchunk = 0; w.zwarm = 0;
  a.walks[k] = w;
}

// One record per alignment, stored AT ITS QUERY ID (the launch runs longest first; the caller's arrays are in id order): what the
// host needs of a finished alignment in one 40-byte line, so that its result loop reads and writes sequentially.
struct BatchRec {
  float score;               // maximum (0: no positive cell; < 0: left undecided by the packed float16 pass)
  uint32_t pos;              // walk: first column of the alignment
  uint32_t len;              // walk: consensus length
  int32_t status;            // walk: 0 ok, 1 left its window, 2 capacity
  int64_t ix, iy;            // argmax cell
  int64_t off;               // walk: offset of its two strings in the consensus buffer
};
static_assert(sizeof(BatchRec) == 40, "one record per alignment");

// ... without walks (score + argmax only): 12 bytes per alignment come down instead of 40
struct BatchRecScore {
  float score;
  int32_t ix, iy;            // (a row of a resident query, a column of a range shorter than 1024: both fit)
};
static_assert(sizeof(BatchRecScore) == 12, "one record per alignment");

__global__ void batch_score_records_by_id(const BatchWaveArgs a, BatchRecScore *rec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.count) return;
  BatchRecScore r;
  r.score = a.sbest[k];
  r.ix = (int32_t)a.scell[2 * (size_t)k]; r.iy = (int32_t)a.scell[2 * (size_t)k + 1];
  rec[a.qsel[batch_sorted_pos(a.first, a.count, k)]] = r;
}

__global__ void batch_records_by_id(const BatchWaveArgs a, const int64_t *wout, const int64_t *offs, BatchRec *rec) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= a.count) return;
  BatchRec r;
  r.score = a.sbest[k];
  r.ix = a.scell[2 * (size_t)k]; r.iy = a.scell[2 * (size_t)k + 1];
  r.pos = 0; r.len = 0; r.status = 0; r.off = 0;
  if (wout != nullptr) {
    r.len = (uint32_t)wout[3 * (size_t)k]; r.pos = (uint32_t)wout[3 * (size_t)k + 1]; r.status = (int32_t)wout[3 * (size_t)k + 2];
    r.off = offs[k];
  }
  rec[a.qsel[batch_sorted_pos(a.first, a.count, k)]] = r;
}

// bytes of consensus the write pass will emit per walk (x and y back to back), 0 for a failed walk
__global__ void batch_walk_sizes(const int64_t *wout, int n, int64_t *sizes) {
  const int k = blockIdx.x * blockDim.x + threadIdx.x;
  if (k >= n) return;
  sizes[k] = wout[3 * (size_t)k + 2] == 0 ? 2 * wout[3 * (size_t)k] : 0;
}

// ---- exclusive scan of int64: v[k] <- sum of v[0..k), total in *total ---------------------------------------------
constexpr int kScanBlock = 256;
constexpr int kScanItems = 16;                                     // per thread: 4096 values per workgroup

__device__ inline int64_t scan_block_exclusive(int64_t v, int64_t *shared, int64_t *block_total) {
  // exclusive prefix of one value per thread over the workgroup (Hillis-Steele over wavefront totals)
  const int lane = threadIdx.x & 63, wv = threadIdx.x >> 6;
  int64_t incl = v;
#pragma unroll
  for (int off = 1; off < 64; off <<= 1) {
    const int64_t o = __shfl_up(incl, off);
    if (lane >= off) incl += o;
  }
  if (lane == 63) shared[wv] = incl;
  __syncthreads();
  int64_t base = 0, tot = 0;
  for (int k = 0; k < kScanBlock / 64; ++k) { if (k < wv) base += shared[k]; tot += shared[k]; }
  __syncthreads();
  *block_total = tot;
  return base + incl - v;
}

__global__ __launch_bounds__(kScanBlock) void scan_partials(const int64_t *v, int64_t n, int64_t *partial) {
  __shared__ int64_t sh[kScanBlock / 64];
  const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) if (base + k < n) s += v[base + k];
  int64_t tot;
  (void)scan_block_exclusive(s, sh, &tot);
  if (threadIdx.x == 0) partial[blockIdx.x] = tot;
}

// one workgroup: exclusive scan of the partials in place (any count), grand total out
__global__ __launch_bounds__(kScanBlock) void scan_of_partials(int64_t *partial, int np, int64_t *total) {
  __shared__ int64_t sh[kScanBlock / 64];
  __shared__ int64_t carry;
  if (threadIdx.x == 0) carry = 0;
  __syncthreads();
  for (int b = 0; b < np; b += kScanBlock) {
    const int k = b + threadIdx.x;
    const int64_t v = k < np ? partial[k] : 0;
    int64_t tot;
    const int64_t ex = scan_block_exclusive(v, sh, &tot);
    const int64_t c = carry;
    if (k < np) partial[k] = c + ex;
    __syncthreads();
    if (threadIdx.x == 0) carry = c + tot;
    __syncthreads();
  }
  if (threadIdx.x == 0) *total = carry;
}

__global__ __launch_bounds__(kScanBlock) void scan_apply(int64_t *v, int64_t n, const int64_t *partial) {
  __shared__ int64_t sh[kScanBlock / 64];
  const int64_t base = ((int64_t)blockIdx.x * kScanBlock + threadIdx.x) * kScanItems;
  int64_t x[kScanItems];
  int64_t s = 0;
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) { x[k] = base + k < n ? v[base + k] : 0; s += x[k]; }
  int64_t tot;
  int64_t run = partial[blockIdx.x] + scan_block_exclusive(s, sh, &tot);
#pragma unroll
  for (int k = 0; k < kScanItems; ++k) {
    if (base + k < n) v[base + k] = run;
    run += x[k];
  }
}

}  // namespace mi355sw
