// roi_align_record_layout.h -- layout of the per-RoI records roi_align_prepare (roi_align_records.hip) leaves in the
// caller's workspace; shared by the NCHW forward / backward kernels and the NHWC forward (roi_align_nhwc.hip).
#pragma once

namespace mi {
namespace {

constexpr int kMaxS = 32;                  // samples per axis on the fast path
constexpr int kMaxStages = 32;             // == max aligned_height on the fast path

// ---- per-RoI record (dwords), stored at the RoI's rank along the sweep ---------------------------------------------
constexpr int kRecHeader = 24;  // [0] flags [1] batch_ind [2] wx0 [3] ww [4] 2^20 / ((ww + 3) / 4) + 1 (divides a lane index by the 16-byte
                                // groups per window row) [5] nstages [6] gh [7] gw [8] roi
                                // [9] wy0 [10] wy1 (last window row) [11] level [12..15] stage 0
                                // [16..17] address of channel 0 of the RoI's image in its level's map (forward calls)
                                // [18] height [19] width of that map [20..23] spare
constexpr int kRecStages = kRecHeader;                 // kMaxStages x {ph0 | ph1 << 16, row0, nrows, 0}
constexpr int kRecY = kRecStages + 4 * kMaxStages;     // kMaxS x {row_lo * ((ww + 3) & ~3) * 4 (the NCHW forward's LDS row
                                                       // pitch), hw / count, lw / count, row_lo}
constexpr int kRecX = kRecY + 4 * kMaxS;               // kMaxS x {(col_lo - wx0) * 4, hw, lw, col_lo}
constexpr int kMaxWin = 63;                            // window rows / columns the backward tables cover
// backward block: the weights of the tile kernel's two passes, merged per window column / row by roi_align_prepare.
//   column c of the window receives, from output column pw, the weight  Wx = sum of the x samples of bin pw that tap c
//   (hw of the samples whose lower tap is c, lw of those whose lower tap is c - 1); rows likewise with hw / count,
//   lw / count.  Entries of one column are contiguous: cf[c] .. cf[c + 1].  At most window + 2 * bins <= 125 entries.
constexpr int kBwdEnt = 128;
constexpr int kRecB = kRecX + 4 * kMaxS;               // dwords: wx[128] f32 | wy[128] f32 | then bytes, see below
constexpr int kBwdWx = 0, kBwdWy = kBwdEnt * 4;        // byte offsets inside the block
constexpr int kBwdPx = 2 * kBwdEnt * 4;                // u8 pw of entry k
constexpr int kBwdPy = kBwdPx + kBwdEnt;               // u8 ph of entry k
constexpr int kBwdCf = kBwdPy + kBwdEnt;               // u8 cf[64]: first entry of window column c (cf[ww] = number of entries)
constexpr int kBwdRf = kBwdCf + 64;                    // u8 rf[64]: first entry of window row r
constexpr int kBwdTabDw = (kBwdRf + 64) / 4;           // 352 dwords = 1408 B
constexpr int kRecDwords = kRecB + kBwdTabDw;          // 760 dwords = 3040 B
static_assert(kRecDwords % 4 == 0 && kRecB % 4 == 0, "records and their backward block are fetched in 16-byte pieces");
// after the records: one int4 per rank {x0, x1 | flag, batch*H + y0, batch*H + y1} = the window columns / "global" rows the
// backward of the RoI can touch ({0x3fffffff, 0, ..}: a RoI of no image), read by the backward to find the RoIs of a tile;
// flag kBoundsNoTables: the record has no backward tables (kFlagBwd clear), the RoI takes the per-sample path
constexpr int kBoundsNoTables = 1 << 30;
// counters in front of the records, zeroed by roi_align_prepare.  Every live counter has a 128-byte line to itself: device-
// scope atomics on one line serialise at ~11 ns apiece whichever word they hit (round 5: the eight ticket words of a resident
// forward in one line cost 20 us per call).
constexpr int kCounterDwords = 1024;
constexpr int kBwdClasses = 6;                         // planned backward: cost classes of the tile entries
constexpr int kBwdCounterStride = 32;
constexpr int kBwdBucket = 512;                        // [kBwdBucket + 256 s + 32 c]: entries filed in class c of counter SET s
                                                       // (c == kBwdClasses: the slice budget used).  Two sets, so that no launch
                                                       // has to zero them: a backward files into set P = ws[kBwdParity], its
                                                       // plan kernel's first workgroup zeroes the OTHER set and publishes P in
                                                       // ws[kBwdInUse]; the tile kernel reads set ws[kBwdInUse] and its first
                                                       // workgroup flips ws[kBwdParity] -- each word is read by all workgroups
                                                       // of one launch and written by one workgroup of the other.
constexpr int kBwdSetStride = 256;
constexpr int kBwdParity = kBwdBucket + 7 * 32;        // 736
constexpr int kBwdInUse = kBwdBucket + kBwdSetStride + 7 * 32;  // 992
static_assert(kBwdClasses + 1 <= 7 && kBwdInUse + 32 <= kCounterDwords, "counter block layout");
constexpr int kNoItem = 0x7fffffff;

// forward LDS path / no such image / backward tile path / y and x tables valid
enum : int { kFlagFast = 1, kFlagZero = 2, kFlagBwd = 4, kFlagTabs = 8 };

}  // namespace
}  // namespace mi
