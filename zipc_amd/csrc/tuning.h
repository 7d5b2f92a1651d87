// tuning.h -- every switch the library reads from the environment, in ONE place.
//
// None of them changes a result: each either picks between two exact implementations of the same step (so that the
// suite can compare BOTH with the oracle: tests/test_gpu_fuzz.py runs the randomized parity loop once per setting, in
// a process of its own) or sizes something for an experiment (tools/).  They are read once per process, the first time
// zd::tuning() is called; the defaults are what measured faster (DESIGN.md section 6 has the numbers).
// Switches whose experiment lost and that nothing uses any more are gone: ZIPC_HIP_CHECKSUM_QUEUES (the two checksum
// passes of a large buffer on one queue), ZIPC_HIP_INFLATE_EXPLORE, ZIPC_HIP_INFLATE_BLOCKS_TRACE.
#pragma once

#include <stddef.h>
#include <stdint.h>

namespace zd {

struct Tuning {
  // ---- deflate: which exact form of a step runs
  bool chain_peel;             // ZIPC_HIP_CHAIN=peel     hash chains by the kernel that orders equal hashes itself (lz_chain_kernel)
                               //                          instead of ordered LDS exchange (lz_chain_xchg_kernel, the default where the
                               //                          context's probe passes)
  long chain_check;            // ZIPC_HIP_CHAIN_CHECK=N   the first N streams of a context's first deflate batch are chained by BOTH kernels and
                               //                          compared, under that batch's load (default 32, at most 16 MiB of source; 0: never).  A
                               //                          difference fails the batch (ZIPC_HIP_ERR_HIP) and moves the context to the ordering kernel
  long parse_segments;         // ZIPC_HIP_PARSE_SEGMENTS  -1 (default): parse and blocks by many waves for few long streams; 0 never;
                               //                          1 whenever a stream has two segments
  long parse_seg;              // ZIPC_HIP_PARSE_SEG       positions per parse segment (default 0: by stream length, 4096-16384)
  long match_tiles_per_group;  // ZIPC_HIP_MATCH_TILES_PER_GROUP  consecutive tiles per lz_match workgroup (default 0: by the grid)
  int match_form;              // ZIPC_HIP_MATCH_FORM      0 (default): lz_match's walk chosen per tile; 1 / 2: always the first / second form
  size_t deflate_group_bytes;  // ZIPC_HIP_DEFLATE_GROUP_BYTES  source bytes per pass through the scratch (default 8 GiB; tests: a few streams)
  long slices, slice_min;      // ZIPC_HIP_SLICES, ZIPC_HIP_SLICE_MIN  a batch cut into slices on side queues (default 0: two slices of at
                               //                          least 2048 streams each, api.hip batch_slices; tests force more and smaller ones)
  // ---- inflate of one long stream by blocks (api.hip inflate_by_blocks)
  bool inflate_blocks;         // ZIPC_HIP_INFLATE_BLOCKS=0  the stream's one wave instead
  int inflate_follow;          // ZIPC_HIP_INFLATE_FOLLOW  -1 (default): sources followed inside the token run in calls of 32 MiB of output and more; 0 / 1 never / always
  uint64_t explore_stride;     // ZIPC_HIP_EXPLORE_STRIDE  input bytes between two explorers (default 16384: an explorer that ends on a false end-of-block starts again, inflate.hip)
  int resolve_hops0, resolve_hops1;  // ZIPC_HIP_RESOLVE_HOPS0 / 1  links a thread follows in the first / a later resolve round (default 256)
  // ---- checksums
  bool checksum_fused;         // ZIPC_HIP_CHECKSUM_FUSED=0  both checksums of one buffer by two passes instead of one
  // ---- host forms (api.hip many_streams)
  long host_threads, host_chunks;  // ZIPC_HIP_HOST_THREADS / ZIPC_HIP_HOST_CHUNKS  staging threads and sub-batches of the many-stream host forms
                               //                          (default 0: 8 threads or the core count; 4 sub-batches, 6 from a GiB staged)
  long host_chunk_min;         // ZIPC_HIP_HOST_CHUNK_MIN  fewest streams a sub-batch of those forms holds (default 1024)
  bool host_pack;              // ZIPC_HIP_HOST_PACK=0     a sub-batch's whole destination slots come back by the copy engine instead of
                               //                          its outputs end to end by a kernel that writes the pinned memory
  long host_pack_wgs;          // ZIPC_HIP_HOST_PACK_WGS   workgroups of that kernel (default 6)
  long host_h2d_mib;           // ZIPC_HIP_HOST_H2D_MIB    a sub-batch's sources go to the device in copies of about this size as they are
                               //                          gathered (default 16; 0: one copy per sub-batch)
  bool host_timing;            // ZIPC_HIP_HOST_TIMING=1   those forms print where each sub-batch was when on stderr (tools/gpu_host_check.sh)
};

const Tuning &tuning();  // api.hip

}  // namespace zd
