// kernels.h -- declarations of the HIP kernels launched by api.hip.
#pragma once

#include "zd_common.h"

namespace zd {

// ---- inflate.hip
constexpr int INFLATE_LDS_BYTES_PER_LANE = 10072;  // = LDS_BYTES_PER_LANE (inflate_lane.h)
constexpr size_t INFLATE_SCRATCH_PER_STREAM = 64 * 18 * 2;  // the span decoder's index (inflate_span.h)
__global__ void inflate_batch_kernel(const uint8_t *__restrict__ src_arena,
                                     uint8_t *__restrict__ dst_arena,
                                     const StreamDesc *__restrict__ descs,
                                     StreamResult *__restrict__ results, uint32_t n_streams,
                                     uint16_t *__restrict__ span_scratch, int crc_op);

__global__ void inflate_batch_few_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                         const StreamDesc *__restrict__ descs, StreamResult *__restrict__ results, uint32_t n_streams,
                                         uint16_t *__restrict__ span_scratch, int crc_op);
// one stream by a wave per block (inflate.hip has the description; api.hip inflate_by_blocks the order of the launches)
struct BlockStart {
  uint64_t bit;       // of the block's header in the stream's input
  uint32_t out_pos;   // of its first byte in the stream's output (the token run)
  uint32_t chunk0;    // the chain: Adler-32 chunks of the blocks before
};
struct BlockEnd {
  uint32_t status, final_block;
  uint64_t end_bit;   // of the first bit behind the block (a stored block: behind its bytes)
  uint32_t out_len, pad;
};
struct BlockRec { uint64_t bit; BlockEnd e; };  // a block that was walked from its header's bit to its end (e.pad: its checkpoints)
// checkpoints of a block's dry run (inflate_span.h SpanCk): bit from the header's bit, output byte from the block's first
constexpr uint32_t BLOCK_CK_MAX = 31;
struct BlockCk { uint32_t n; uint32_t e[2 * BLOCK_CK_MAX]; uint32_t pad; };
struct ChainIv { uint32_t first, ck; };  // a chain block's first interval (the token run: a wave per interval), its BlockCk
constexpr int RESOLVE_ROUNDS = 12;  // (h hops a round: pointers of h^r copies after round r)
struct FindCounts {
  uint32_t n_first;   // offsets that passed the header test (may exceed the list: those are lost)
  uint32_t n_cand;    // candidates (likewise)
  uint32_t chain_ok;  // inflate_chain_kernel: 1 = the blocks chain up to a final one and fit
  uint32_t n_blocks;
  uint64_t out_len;
  uint32_t token_bad; // inflate_blocks_token_kernel: blocks that did not end as the dry run said
  uint32_t more[RESOLVE_ROUNDS];  // inflate_resolve_kernel: bytes round r left short of a literal
  uint32_t n_walked;  // inflate_chain_kernel: blocks of the chain that it had to walk itself
  uint32_t n_recs;    // blocks listed: the candidates' (inflate_blocks_dry_kernel), then the explorers' (may exceed the list)
  uint32_t n_chunks;  // inflate_chain_kernel: Adler-32 chunks of the chain's blocks (every block has its own grid, zd.ml:682-690)
  uint32_t n_intervals, pad2;  // inflate_chain_kernel: intervals of the chain's blocks (a block and its checkpoints)
  uint64_t miss_bit;  // inflate_chain_kernel without walking: where the chain could not go on (~0: nowhere)
};
// A stream of a call that goes by blocks: where its lists live and what this launch takes of it.  The kernels' grids
// have the call's streams as their second dimension (jobs[blockIdx.y]) and the longest stream's need as their first.
struct BlocksJob {
  uint32_t stream;           // its descriptor and result
  uint32_t first_cap, cand_cap, rec_cap, chain_cap;
  uint32_t n;                // this launch's waves of the stream: explorers, or intervals / blocks of the token run
  uint32_t n_blocks;         // blocks of its chain (the explore launch: how many of its n waves are explorers, the rest followers)
  uint32_t out_len;          // its output bytes
  int32_t follow, pad;
  FindCounts *counts;
  uint32_t *first, *cand;
  BlockRec *recs, *sorted;
  uint32_t *sorted_src;
  BlockStart *chain;
  BlockEnd *chain_end;
  ChainIv *chain_iv;
  BlockCk *cks;
  uint16_t *span;            // the span decoder's index, a slot per wave of the launch
  uint32_t *tok;             // a word per output byte, then the two lists of the resolve rounds
  uint32_t *sums;            // Adler-32: three words per chunk
};
__global__ void inflate_find_headers_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                            const BlocksJob *__restrict__ jobs);
__global__ void inflate_find_lengths_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                            const BlocksJob *__restrict__ jobs);
__global__ void inflate_blocks_dry_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                          const StreamDesc *__restrict__ descs, const BlocksJob *__restrict__ jobs);
__global__ void inflate_explore_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                       const StreamDesc *__restrict__ descs, const BlocksJob *__restrict__ jobs, uint32_t stride_bits);
__global__ void inflate_sort_blocks_kernel(const BlocksJob *__restrict__ jobs);
__global__ void inflate_chain_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                     const StreamDesc *__restrict__ descs, const BlocksJob *__restrict__ jobs, int walk);
__global__ void inflate_tok_init_kernel(const BlocksJob *__restrict__ jobs);
__global__ void inflate_blocks_token_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                            const StreamDesc *__restrict__ descs, const BlocksJob *__restrict__ jobs);
__global__ void inflate_resolve_kernel(const BlocksJob *__restrict__ jobs, int round, int hops);
__global__ void inflate_adler_chunks_kernel(const uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
                                            const BlocksJob *__restrict__ jobs);
__global__ void inflate_adler_fold_kernel(const BlocksJob *__restrict__ jobs, int rfc, StreamResult *__restrict__ results);
__global__ void inflate_blocks_result_kernel(const BlocksJob *__restrict__ jobs, StreamResult *__restrict__ results, uint32_t n_jobs);
__global__ void inflate_gather_kernel(uint8_t *__restrict__ dst_arena, const StreamDesc *__restrict__ descs,
                                      const BlocksJob *__restrict__ jobs);

// a huge stream of equal stored blocks (inflate.hip, api.hip)
struct StoredChain {
  uint32_t len0;        // LEN of the first block (0: the stream does not start with a stored block)
  uint32_t candidates;  // header positions j * (5 + len0) inside the input
  uint32_t first_bad;   // first candidate that is not a stored header of that length
  uint32_t final_at;    // first candidate with BFINAL set (0xFFFFFFFF: none)
};
__global__ void stored_chain_probe_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                          StoredChain *__restrict__ st);
__global__ void stored_chain_scan_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                         StoredChain *__restrict__ st);
__global__ void stored_chain_copy_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                         const StreamDesc *__restrict__ descs, uint32_t len0);
// ... and of stored blocks of ANY lengths: one wave walks the headers (64 at a time while the blocks keep
// their length) and lists the blocks; a second kernel copies the listed blocks
struct StoredBlock { uint64_t src, dst; uint32_t len, pad; };  // offsets inside the stream's source / destination
enum : uint32_t { WALK_MORE = 0, WALK_FINAL = 1, WALK_OTHER = 2, WALK_CORRUPT = 3, WALK_ROOM = 4 };
struct StoredWalk {
  uint64_t src_pos, dst_pos;  // in: where the walk starts; out: where it stopped (a header's first byte)
  uint64_t room;              // output bytes the walk may still list
  uint32_t n_blocks;          // out: blocks listed
  uint32_t stop;              // out: WALK_* -- the list is full / the final block is listed / the next block is not a
                              // stored one / its header is damaged or cut short (the reference's "Corrupted data
                              // stream", zd.ml:672-677) / the next block does not fit the room
};
__global__ void stored_walk_kernel(const uint8_t *__restrict__ src_arena, const StreamDesc *__restrict__ descs,
                                   StoredWalk *__restrict__ walk, StoredBlock *__restrict__ list, uint32_t list_cap);
__global__ void stored_list_copy_kernel(const uint8_t *__restrict__ src_arena, uint8_t *__restrict__ dst_arena,
                                        const StreamDesc *__restrict__ descs, const StoredBlock *__restrict__ list,
                                        uint32_t n_blocks);

// ---- checksum.hip
constexpr uint32_t CRC_PIECE_BYTES = 128;  // bytes per thread of crc32_segments_kernel
constexpr uint32_t CRC_SEG_BYTES = 32768;   // per workgroup (256 threads)
enum : int { RANGE_INFLATE_OUT = 0, RANGE_DEFLATE_SRC = 1, RANGE_SINGLE = 2 };
struct CrcConsts {
  uint32_t xpiece[8];
  uint32_t xseg;
  uint32_t xbyte[48];  // x^(8 * 2^k) mod P: x^(8n) is one multiply per set bit of n
};
// nibble tables (zd_common.h gf2_mul_nib) of the constant multipliers, in device memory
// owned by the context: xpiece[0..7], then xseg
constexpr int CRC_NIB_CONSTS = 9;
constexpr int CRC_NIB_XSEG = 8;
__global__ void crc32_segments_kernel(const uint8_t *__restrict__ base, int mode,
                                      const StreamDesc *__restrict__ descs,
                                      const StreamResult *__restrict__ results,
                                      uint64_t single_off, uint64_t single_len,
                                      uint32_t segs_per_range, const uint32_t *__restrict__ nib,
                                      uint32_t *__restrict__ partials);
__global__ void crc32_adler_segments_kernel(const uint8_t *__restrict__ p, uint64_t len, uint32_t n_segs,
                                            const uint32_t *__restrict__ nib, uint32_t *__restrict__ partials,
                                            uint2 *__restrict__ adler_sums, uint64_t n_chunks);
__global__ void crc32_finish_kernel(int mode, const StreamDesc *__restrict__ descs,
                                    StreamResult *__restrict__ results, uint64_t single_len,
                                    uint32_t segs_per_range, CrcConsts K, const uint32_t *__restrict__ nib,
                                    const uint32_t *__restrict__ partials,
                                    uint32_t *__restrict__ single_out);
__global__ void crc32_finish_streams_kernel(int mode, const StreamDesc *__restrict__ descs,
                                            StreamResult *__restrict__ results, uint32_t n_ranges,
                                            uint32_t segs_per_range, CrcConsts K,
                                            const uint32_t *__restrict__ partials);
__global__ void adler_chunks_kernel(const uint8_t *__restrict__ p, uint64_t n, uint64_t n_chunks,
                                    uint2 *__restrict__ sums);
// RFC 1950's Adler-32 from the chunk sums (one workgroup)
__global__ void adler_rfc_finish_kernel(const uint2 *__restrict__ sums, uint64_t n, uint64_t n_chunks,
                                        uint32_t *__restrict__ out);
constexpr uint32_t ADLER_AMB_CAP = 8192;  // ambiguous-chunk records (16 bytes each)
constexpr uint32_t ADLER_MAX_RUNS = 65536;
// per-run arrays of the Adler chain (device scratch, n_runs entries each)
struct AdlerRuns {
  uint32_t n_runs;       // a multiple of 1024
  uint32_t *sum;         // S1 sum of the run, later its a sum (mod p)
  uint32_t *s1_before;   // s1 before the run
  uint32_t *s1_after;
  uint32_t *last_hi;     // branch of the run's last chunk, 0xFFFFFFFF for an empty run
  uint32_t *res_before;  // residue of s2 before the run
  uint32_t *amb_count;   // [1]
};
__global__ void adler_runs_s1_kernel(const uint2 *__restrict__ sums, uint64_t n_chunks, uint64_t per, AdlerRuns R);
__global__ void adler_scan_runs_kernel(const uint32_t *__restrict__ in, uint32_t *__restrict__ out, uint32_t n_runs,
                                       uint32_t first);
__global__ void adler_runs_a_kernel(const uint2 *__restrict__ sums, uint64_t n, uint64_t n_chunks, uint64_t per,
                                    AdlerRuns R, uint32_t *__restrict__ amb, uint32_t amb_cap);
__global__ void adler_replay_kernel(const uint2 *__restrict__ sums, uint64_t n, uint64_t n_chunks, uint64_t per,
                                    AdlerRuns R, uint32_t *__restrict__ amb, uint32_t amb_cap,
                                    uint32_t *__restrict__ out);

}  // namespace zd
