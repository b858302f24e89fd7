// fx_blocks.hip.h -- a channel's sample stream as fx_push_samples sees it: [pending samples | new block], two pieces of memory read as
// one run of bytes.  ref AudioDataCollector.h:36-94: the audio callback writes blocks of the DEVICE's length into the collector's ring and
// the analysis thread reads window/2 samples from wherever the read index stands; here the "ring" is the channel's pending samples
// (fewer than window/2, the `carry`: a 16-byte aligned row) followed by the block just delivered (a row of the caller's buffer, at
// whatever alignment the block length gives it).
//
// Used by fx_reblock.hip (the byte mover behind the calls the kernels do not read themselves) and by the kernels' load stage (BLOCKS forms
// of fx_hop_kernel / fx_frame_kernel<direct> / fx_frame_tail_kernel, one launch per hop of a call that completes one or two -- the live
// case: 441 / 480 / 512 / 960 / 1024-sample device blocks against hops of 512 .. 2048 -- and, at 1024 points, of the batch kernel for calls of
// any length): the window is read straight from the two pieces and only what is left over is written, so a sample crosses HBM once on
// its way in instead of three times.
// Included inside namespace fxk; not a stand-alone header.

struct BlockStream {
    const unsigned char* carry_row;   // the channel's pending samples, carry_bytes of them valid; 16-byte aligned
    const unsigned char* block_row;   // the channel's row of the new block
    int                  carry_bytes;
    long long            in_row_bytes;
};

// dwords that need only dword alignment (the hardware's requirement for a multi-dword global access)
struct __attribute__((packed, aligned(4))) dwords4 { unsigned x, y, z, w; };
struct __attribute__((packed, aligned(4))) dwords3 { unsigned x, y, z; };
struct __attribute__((packed, aligned(4))) dwords2 { unsigned x, y; };

// the same stream from byte `shift` on (a multiple of 16: whole hops): wave-uniform arithmetic, so a launch that analyses hop k of a block
// pays for it in scalar registers, not in every lane's addresses
__device__ __forceinline__ BlockStream stream_from(const BlockStream& s, long long shift)
{
    BlockStream r = s;
    if (shift < s.carry_bytes) { r.carry_row += shift; r.carry_bytes -= (int) shift; }
    else { r.block_row += shift - s.carry_bytes; r.in_row_bytes -= shift - s.carry_bytes; r.carry_bytes = 0; }
    return r;
}

// byte s of the stream (s < carry_bytes + in_row_bytes)
__device__ __forceinline__ unsigned stream_byte(const BlockStream& s, long long at)
{
    return at < s.carry_bytes ? s.carry_row[at] : s.block_row[at - s.carry_bytes];
}

// One 16-byte piece of the stream at byte offset d0 (a multiple of 16), zeros past `total`.
__device__ __forceinline__ void stream_piece16(const BlockStream& s, long long total, long long d0, unsigned (&v)[4])
{
    const long long b0 = d0 - s.carry_bytes;
    if (b0 >= 0 && b0 + 20 <= s.in_row_bytes) {
        // wholly inside the new block, and so are the five aligned dwords around it
        // (pointer arithmetic, not an integer round trip: the compiler keeps the global address space and emits global_load_dwordx4)
        const unsigned char* at = s.block_row + b0;
        const unsigned sh = (unsigned) (reinterpret_cast<uintptr_t>(at) & 3);
        const unsigned char* base = at - sh;
        const dwords4 q = *reinterpret_cast<const dwords4*>(base);
        const unsigned q4 = *reinterpret_cast<const unsigned*>(base + 16);
        v[0] = __builtin_amdgcn_alignbyte(q.y, q.x, sh);
        v[1] = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
        v[2] = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
        v[3] = __builtin_amdgcn_alignbyte(q4, q.w, sh);
    } else if (d0 + 16 <= s.carry_bytes) {
        // wholly inside the pending samples: the carry row starts on a 16-byte boundary and so does this piece
        const uint4 q = *reinterpret_cast<const uint4*>(s.carry_row + d0);
        v[0] = q.x; v[1] = q.y; v[2] = q.z; v[3] = q.w;
    } else if (b0 >= 0) {
        // the block's last bytes (one piece per row): the same five dwords, each one read only as far as the row goes -- whole where it
        // lies inside the row, its leading bytes where the row ends inside it, nothing beyond (what is not read is zero, as the bytes past
        // `total` must be)
        const unsigned char* at = s.block_row + b0;
        const unsigned sh = (unsigned) (reinterpret_cast<uintptr_t>(at) & 3);
        const unsigned char* base = at - sh;
        const unsigned char* end = s.block_row + s.in_row_bytes;
        unsigned q[5];
#pragma unroll
        for (int j = 0; j < 5; j++) {
            const unsigned char* w = base + 4 * j;
            q[j] = 0;
            if (w + 4 <= end) q[j] = *reinterpret_cast<const unsigned*>(w);
            else if (w < end) {
#pragma unroll
                for (int k = 0; k < 3; k++) if (w + k < end) q[j] |= (unsigned) w[k] << (8 * k);
            }
        }
#pragma unroll
        for (int j = 0; j < 4; j++) v[j] = __builtin_amdgcn_alignbyte(q[j + 1], q[j], sh);
    } else {
        // across the boundary between the pending samples and the block (pending counts that are no multiple of 16 bytes): byte by byte
#pragma unroll
        for (int j = 0; j < 4; j++) {
            v[j] = 0;
#pragma unroll
            for (int k = 0; k < 4; k++) {
                const long long at = d0 + 4 * j + k;
                if (at < total) v[j] |= stream_byte(s, at) << (8 * k);
            }
        }
    }
}

// The 4 * BYTES bytes of four consecutive samples of BYTES bytes each, starting at sample i (a multiple of 4) of the stream; all of
// them inside the stream.  Returned as fetch_four returns them (fx_frame_kernel.hip.h): the first 4 * BYTES bytes of the uint4.
template <int BYTES> __device__ __forceinline__ uint4 stream_four(const BlockStream& s, int i)
{
    constexpr int W = 4 * BYTES;                       // 16, 12 or 8
    const long long d0 = (long long) i * BYTES;
    uint4 r = uint4{0u, 0u, 0u, 0u};
    if (d0 + W <= s.carry_bytes) {
        // pending samples: W bytes from a multiple of W in a 16-byte aligned row
        const unsigned char* at = s.carry_row + d0;
        if constexpr (BYTES == 4) r = *reinterpret_cast<const uint4*>(at);
        else if constexpr (BYTES == 3) { const dwords3 q = *reinterpret_cast<const dwords3*>(at); r.x = q.x; r.y = q.y; r.z = q.z; }
        else { const uint2 q = *reinterpret_cast<const uint2*>(at); r.x = q.x; r.y = q.y; }
        return r;
    }
    const long long b0 = d0 - s.carry_bytes;
    if (b0 >= 0 && b0 + W + 4 <= s.in_row_bytes) {
        // the new block, at whatever byte the block length left this row and the pending count left this sample on: the aligned dwords
        // around the bytes, shifted into place
        const unsigned char* at = s.block_row + b0;
        const unsigned sh = (unsigned) (reinterpret_cast<uintptr_t>(at) & 3);
        const unsigned char* base = at - sh;
        if constexpr (BYTES == 4) {
            // (four-byte samples: the block starts on a dword -- the API asks for that -- and rows and pending counts are whole samples)
            const dwords4 q = *reinterpret_cast<const dwords4*>(at);
            r.x = q.x; r.y = q.y; r.z = q.z; r.w = q.w;
        } else if constexpr (BYTES == 3) {
            const dwords4 q = *reinterpret_cast<const dwords4*>(base);
            r.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh); r.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh); r.z = __builtin_amdgcn_alignbyte(q.w, q.z, sh);
        } else {
            const dwords3 q = *reinterpret_cast<const dwords3*>(base);
            r.x = __builtin_amdgcn_alignbyte(q.y, q.x, sh); r.y = __builtin_amdgcn_alignbyte(q.z, q.y, sh);
        }
        return r;
    }
    // across the boundary between the two pieces, or the block's last bytes: byte by byte
    unsigned v[4] = {0u, 0u, 0u, 0u};
#pragma unroll
    for (int k = 0; k < W; k++) v[k >> 2] |= stream_byte(s, d0 + k) << (8 * (k & 3));
    return uint4{v[0], v[1], v[2], v[3]};
}

// where sample idx (of BYTES bytes) of the stream lives; the sample does not straddle the two pieces (both hold whole samples)
template <int BYTES> __device__ __forceinline__ const unsigned char* stream_sample(const BlockStream& s, int idx)
{
    const long long d0 = (long long) idx * BYTES;
    return d0 < s.carry_bytes ? s.carry_row + d0 : s.block_row + (d0 - s.carry_bytes);
}

// What a one-hop call leaves over: bytes [hop_bytes, total) of the stream become the channel's new pending samples (16-byte pieces, the
// last one zero-filled), written by `threads` threads of which this is number `tid`.  hop_bytes is a multiple of 16.
__device__ __forceinline__ void stream_keep_rest(const BlockStream& s, long long hop_bytes, unsigned char* carry_out_row, int tid, int threads)
{
    const long long total = (long long) s.carry_bytes + s.in_row_bytes;
    for (long long d = hop_bytes + 16ll * tid; d < total; d += 16ll * threads) {
        unsigned v[4];
        stream_piece16(s, total, d, v);
        *reinterpret_cast<uint4*>(carry_out_row + (d - hop_bytes)) = uint4{v[0], v[1], v[2], v[3]};
    }
}

// What the host must have made sure of before a block-fed launch (FrameParams::block_mode): the kernels index on trust.
static inline bool block_feed_valid(int n, const FrameParams& p)
{
    const long long esz = p.sample_format == FX_SAMPLE_F32 ? 4 : (p.sample_format == FX_SAMPLE_S24 ? 3 : 2);
    const long long hop = (long long) (n / 2) * esz, total = (long long) p.blk_carry_bytes + p.blk_in_row_bytes;
    return n >= 1024 && p.T >= 1 && p.T <= 4096 && p.hop_mode == 1 && p.in && p.blk_carry_in && p.blk_carry_out && p.blk_carry_in != p.blk_carry_out &&
           p.blk_carry_bytes >= 0 && p.blk_carry_bytes < hop && p.blk_carry_bytes % esz == 0 && p.blk_in_row_bytes > 0 && p.blk_in_row_bytes % esz == 0 &&
           p.blk_hop0 >= 0 && p.blk_hop0 < 4096 && total >= (p.blk_hop0 + p.T) * hop && (!p.blk_keep_rest || total < (p.blk_hop0 + p.T + 1) * hop) &&
           (p.blk_carry_row_bytes & 15) == 0 && p.blk_carry_row_bytes >= (n / 2) * 4 &&
           (reinterpret_cast<uintptr_t>(p.in) & 3) == 0 && (reinterpret_cast<uintptr_t>(p.blk_carry_in) & 15) == 0 && (reinterpret_cast<uintptr_t>(p.blk_carry_out) & 15) == 0;
}

