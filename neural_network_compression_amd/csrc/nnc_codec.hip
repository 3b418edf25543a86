// nnc_codec.hip -- the compressed form of a quantized layer: Huffman-coded centroid indices.
//
// The reference stops at `cluster_centers_[labels_]` (neural_network_compression/common/utility.py:239) and stores
// nothing; its README names the third stage of Deep Compression (Huffman coding, README.md:9) without implementing it
// (SURVEY 8f-3).  What the path already produces -- the centroid index of every weight, the codebook, the index histogram
// and the Huffman code lengths (nnc_huffman_lengths) -- is all a stored layer needs; this file adds the bit packing:
//
//   canonical Huffman codes from the code lengths (host, K-sized);
//   encode: chunks of NNC_CODEC_CHUNK indices; pass 1 adds up the code lengths of every chunk, an exclusive scan gives the
//           chunks their bit offsets, pass 2 packs each chunk's codes through LDS and merges them into the word stream
//           (MSB first; only a chunk's first and last word can be shared with a neighbour: atomic OR);
//   decode: one thread per chunk walks its bits with the canonical first-code table (chunks are independent, so the
//           decoder is as parallel as the chunk table is long).
// Pruned weights are exact zeros that all fall in one cluster, so their index is by far the most frequent symbol and gets a
// one-bit code: the dense index stream already is the sparse format (no separate position stream to keep in step).
#include <hip/hip_runtime.h>

#include <algorithm>
#include <string>
#include <vector>

#include "nnc.h"

int nnc_set_error_(int code, const char *msg); // in nnc_hip.hip

#define CODEC_CHUNK NNC_CODEC_CHUNK
#define CODEC_MAXLEN 32

// Canonical codes: symbols ordered by (length, symbol); the first code of length l follows from the counts of shorter ones.
extern "C" int nnc_huffman_codes(const uint8_t *lengths, int32_t k, uint32_t *codes_out)
{
    if (!lengths || !codes_out || k < 1 || k > NNC_KMAX) return nnc_set_error_(NNC_EINVAL, "nnc_huffman_codes: bad argument");
    int count[CODEC_MAXLEN + 2] = {0};
    for (int s = 0; s < k; s++) {
        if (lengths[s] > CODEC_MAXLEN) return nnc_set_error_(NNC_EINVAL, "nnc_huffman_codes: a code is longer than 32 bits (flatten the lengths first)");
        count[lengths[s]]++;
    }
    count[0] = 0;
    uint64_t next[CODEC_MAXLEN + 2] = {0};
    uint64_t code = 0;
    for (int l = 1; l <= CODEC_MAXLEN; l++) {
        code = (code + (uint64_t)count[l - 1]) << 1;
        next[l] = code;
    }
    // Kraft: the codes of the longest length must not run past 2^l
    for (int s = 0; s < k; s++) {
        const int l = lengths[s];
        codes_out[s] = l ? (uint32_t)next[l]++ : 0u;
    }
    for (int l = 1; l <= CODEC_MAXLEN; l++)
        if (count[l] && next[l] > ((uint64_t)1 << l)) return nnc_set_error_(NNC_EINVAL, "nnc_huffman_codes: the lengths violate Kraft's inequality");
    return NNC_OK;
}

template <typename LT>
__global__ __launch_bounds__(256) void k_chunk_bits(const LT *__restrict__ labels, long long n, const uint8_t *__restrict__ lengths, int k,
                                                    unsigned long long *__restrict__ chunk_bits)
{
    __shared__ uint8_t len_s[NNC_KMAX];
    __shared__ unsigned wsum[4];
    for (int i = threadIdx.x; i < k; i += 256) len_s[i] = lengths[i];
    __syncthreads();
    const long long base = (long long)blockIdx.x * CODEC_CHUNK;
    unsigned bits = 0;
    for (int i = threadIdx.x; i < CODEC_CHUNK; i += 256) {
        const long long g = base + i;
        if (g < n) { const int l = (int)labels[g]; bits += l < k ? len_s[l] : 0; }
    }
    for (int off = 32; off >= 1; off >>= 1) bits += __shfl_xor(bits, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = bits;
    __syncthreads();
    if (threadIdx.x == 0) chunk_bits[blockIdx.x] = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

// exclusive scan of nchunks counts in place (+ the total behind them), one workgroup
__global__ __launch_bounds__(1024) void k_scan_bits(unsigned long long *__restrict__ v, long long nchunks)
{
    __shared__ unsigned long long wave_tot[16];
    __shared__ unsigned long long carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    if (tid == 0) carry_s = 0;
    __syncthreads();
    for (long long base = 0; base < nchunks; base += 1024 * 8) {
        unsigned long long a[8];
        const long long i0 = base + (long long)tid * 8;
        for (int u = 0; u < 8; u++) a[u] = (i0 + u < nchunks) ? v[i0 + u] : 0;
        unsigned long long s = 0;
        for (int u = 0; u < 8; u++) s += a[u];
        unsigned long long inc = s;
        for (int off = 1; off < 64; off <<= 1) { const unsigned long long t = __shfl_up(inc, off); if (lane >= off) inc += t; }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        unsigned long long pre = carry_s, tot = 0;
        for (int w = 0; w < 16; w++) { if (w < wv) pre += wave_tot[w]; tot += wave_tot[w]; }
        unsigned long long run = pre + (inc - s);
        for (int u = 0; u < 8; u++) if (i0 + u < nchunks) { v[i0 + u] = run; run += a[u]; }
        __syncthreads();
        if (tid == 0) carry_s += tot;
        __syncthreads();
    }
    if (tid == 0) v[nchunks] = carry_s;
}

template <typename LT>
__global__ __launch_bounds__(256) void k_huff_encode(const LT *__restrict__ labels, long long n, const uint32_t *__restrict__ codes,
                                                     const uint8_t *__restrict__ lengths, int k, const unsigned long long *__restrict__ chunk_off,
                                                     uint32_t *__restrict__ words)
{
    __shared__ uint32_t code_s[NNC_KMAX];
    __shared__ uint8_t len_s[NNC_KMAX];
    __shared__ uint32_t buf[CODEC_CHUNK + 2]; // CODEC_CHUNK symbols * <= 32 bits, plus the lead-in word
    __shared__ unsigned wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    for (int i = tid; i < k; i += 256) { code_s[i] = codes[i]; len_s[i] = lengths[i]; }
    for (int i = tid; i < CODEC_CHUNK + 2; i += 256) buf[i] = 0u;
    __syncthreads();
    const long long base = (long long)blockIdx.x * CODEC_CHUNK;
    const unsigned long long off0 = chunk_off[blockIdx.x], off1 = chunk_off[blockIdx.x + 1];
    const unsigned lead = (unsigned)(off0 & 31ull);
    // a thread takes CODEC_CHUNK / 256 consecutive symbols
    constexpr int PER = CODEC_CHUNK / 256;
    uint32_t c[PER];
    unsigned l[PER], mine = 0;
#pragma unroll
    for (int u = 0; u < PER; u++) {
        const long long g = base + tid * PER + u;
        int s = (g < n) ? (int)labels[g] : -1;
        if (s >= k) s = -1;
        c[u] = s >= 0 ? code_s[s] : 0u;
        l[u] = s >= 0 ? len_s[s] : 0u;
        mine += l[u];
    }
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    unsigned pos = lead + (inc - mine);
    for (int w = 0; w < wv; w++) pos += wave_tot[w];
#pragma unroll
    for (int u = 0; u < PER; u++) {
        if (l[u]) {
            // code bits occupy stream positions pos .. pos + l - 1, MSB first
            const unsigned w0 = pos >> 5, b0 = pos & 31u;
            const unsigned long long v = (unsigned long long)c[u] << (64 - l[u] - b0); // left-aligned in a 64-bit window starting at word w0
            atomicOr(&buf[w0], (uint32_t)(v >> 32));
            if (b0 + l[u] > 32) atomicOr(&buf[w0 + 1], (uint32_t)v);
            pos += l[u];
        }
    }
    __syncthreads();
    const unsigned long long wfirst = off0 >> 5;
    const unsigned nwords = (unsigned)(((off1 + 31ull) >> 5) - wfirst); // words this chunk touches (0 if it is empty)
    for (unsigned i = tid; i < nwords; i += 256) {
        const uint32_t v = buf[i];
        if (i == 0 || i == nwords - 1) { if (v) atomicOr(&words[wfirst + i], v); } // may be shared with a neighbouring chunk
        else words[wfirst + i] = v;
    }
}

// canonical decoding tables in LDS: for every length the first code, the number of codes and where its symbols start
template <typename LT>
__global__ __launch_bounds__(256) void k_huff_decode(const uint32_t *__restrict__ words, const unsigned long long *__restrict__ chunk_off, long long n,
                                                     long long nchunks, const uint32_t *__restrict__ first_code, const uint32_t *__restrict__ count,
                                                     const uint32_t *__restrict__ first_index, const uint16_t *__restrict__ sorted_syms, int k,
                                                     LT *__restrict__ labels_out, int *__restrict__ bad)
{
    __shared__ uint32_t fc[CODEC_MAXLEN + 1], cn[CODEC_MAXLEN + 1], fi[CODEC_MAXLEN + 1];
    __shared__ uint16_t syms[NNC_KMAX];
    if (threadIdx.x <= CODEC_MAXLEN) { fc[threadIdx.x] = first_code[threadIdx.x]; cn[threadIdx.x] = count[threadIdx.x]; fi[threadIdx.x] = first_index[threadIdx.x]; }
    for (int i = threadIdx.x; i < k; i += 256) syms[i] = sorted_syms[i];
    __syncthreads();
    const long long chunk = (long long)blockIdx.x * 256 + threadIdx.x;
    if (chunk >= nchunks) return;
    unsigned long long pos = chunk_off[chunk];
    const unsigned long long end = chunk_off[chunk + 1];
    const long long base = chunk * CODEC_CHUNK;
    const int cnt = (int)((base + CODEC_CHUNK < n ? base + CODEC_CHUNK : n) - base);
    uint32_t cur = words[pos >> 5];
    for (int i = 0; i < cnt; i++) {
        uint32_t code = 0;
        int len = 0, sym = -1;
        while (len < CODEC_MAXLEN && pos < end) {
            const unsigned b = (unsigned)(pos & 31ull);
            code = (code << 1) | ((cur >> (31 - b)) & 1u);
            len++;
            pos++;
            if ((pos & 31ull) == 0) cur = words[pos >> 5];
            const uint32_t d = code - fc[len];
            if (code >= fc[len] && d < cn[len]) { sym = syms[fi[len] + d]; break; }
        }
        if (sym < 0) { *bad = 1; sym = 0; }
        labels_out[base + i] = (LT)sym;
    }
    if (pos != end) *bad = 1;
}

extern "C" size_t nnc_codec_chunks(int64_t n) { return n > 0 ? (size_t)((n + CODEC_CHUNK - 1) / CODEC_CHUNK) : 0; }

// chunk_off_dev: (nchunks + 1) uint64; on return chunk_off_dev[c] = first bit of chunk c, chunk_off_dev[nchunks] = total bits.
extern "C" int nnc_huffman_chunk_offsets(const void *labels, int label_bytes, int64_t n, const uint8_t *lengths_dev, int32_t k,
                                         uint64_t *chunk_off_dev, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || !lengths_dev || !chunk_off_dev || (n > 0 && !labels) || (label_bytes != 1 && label_bytes != 2))
        return nnc_set_error_(NNC_EINVAL, "nnc_huffman_chunk_offsets: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long nchunks = (long long)nnc_codec_chunks(n);
    unsigned long long *co = reinterpret_cast<unsigned long long *>(chunk_off_dev);
    if (nchunks > 0) {
        if (label_bytes == 1) hipLaunchKernelGGL((k_chunk_bits<uint8_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(labels), (long long)n, lengths_dev, (int)k, co);
        else hipLaunchKernelGGL((k_chunk_bits<uint16_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(labels), (long long)n, lengths_dev, (int)k, co);
    }
    hipLaunchKernelGGL(k_scan_bits, dim3(1), dim3(1024), 0, s, co, nchunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

// words_dev: ceil(total_bits / 32) + 1 uint32, zeroed here.
extern "C" int nnc_huffman_encode(const void *labels, int label_bytes, int64_t n, const uint32_t *codes_dev, const uint8_t *lengths_dev, int32_t k,
                                  const uint64_t *chunk_off_dev, uint32_t *words_dev, int64_t nwords, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || !codes_dev || !lengths_dev || !chunk_off_dev || !words_dev || nwords < 1 || (n > 0 && !labels) ||
        (label_bytes != 1 && label_bytes != 2))
        return nnc_set_error_(NNC_EINVAL, "nnc_huffman_encode: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(words_dev, 0, (size_t)nwords * 4, s) != hipSuccess) return nnc_set_error_(NNC_EHIP, "nnc_huffman_encode: memset");
    const long long nchunks = (long long)nnc_codec_chunks(n);
    if (nchunks == 0) return NNC_OK;
    const unsigned long long *co = reinterpret_cast<const unsigned long long *>(chunk_off_dev);
    if (label_bytes == 1) hipLaunchKernelGGL((k_huff_encode<uint8_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(labels), (long long)n, codes_dev, lengths_dev, (int)k, co, words_dev);
    else hipLaunchKernelGGL((k_huff_encode<uint16_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(labels), (long long)n, codes_dev, lengths_dev, (int)k, co, words_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

// tables_dev: 3 * 33 uint32 {first_code[33], count[33], first_index[33]} then k uint16 symbols sorted by (length, symbol),
// as nnc_huffman_decode_tables (host) lays them out.  bad_dev: int32, set to 1 if the stream does not parse.
extern "C" int nnc_huffman_decode_tables(const uint8_t *lengths, int32_t k, void *tables_out, size_t tables_bytes)
{
    const size_t need = 3 * (CODEC_MAXLEN + 1) * 4 + (size_t)NNC_KMAX * 2;
    if (!lengths || k < 1 || k > NNC_KMAX || !tables_out || tables_bytes < need) return nnc_set_error_(NNC_EINVAL, "nnc_huffman_decode_tables: bad argument");
    uint32_t *fc = reinterpret_cast<uint32_t *>(tables_out), *cn = fc + (CODEC_MAXLEN + 1), *fi = cn + (CODEC_MAXLEN + 1);
    uint16_t *syms = reinterpret_cast<uint16_t *>(fi + (CODEC_MAXLEN + 1));
    std::vector<uint32_t> codes((size_t)k);
    int rc = nnc_huffman_codes(lengths, k, codes.data());
    if (rc) return rc;
    for (int l = 0; l <= CODEC_MAXLEN; l++) { fc[l] = 0; cn[l] = 0; fi[l] = 0; }
    for (int s = 0; s < k; s++) cn[lengths[s]]++;
    cn[0] = 0;
    uint32_t idx = 0;
    uint64_t code = 0;
    for (int l = 1; l <= CODEC_MAXLEN; l++) {
        code = (code + (uint64_t)cn[l - 1]) << 1;
        fc[l] = (uint32_t)code;
        fi[l] = idx;
        idx += cn[l];
    }
    std::vector<uint32_t> fill(CODEC_MAXLEN + 1);
    for (int l = 0; l <= CODEC_MAXLEN; l++) fill[l] = fi[l];
    for (int s = 0; s < NNC_KMAX; s++) syms[s] = 0;
    for (int s = 0; s < k; s++) if (lengths[s]) syms[fill[lengths[s]]++] = (uint16_t)s;
    return NNC_OK;
}

extern "C" size_t nnc_huffman_decode_tables_bytes(void) { return 3 * (CODEC_MAXLEN + 1) * 4 + (size_t)NNC_KMAX * 2; }

extern "C" int nnc_huffman_decode(const uint32_t *words_dev, const uint64_t *chunk_off_dev, int64_t n, const void *tables_dev, int32_t k,
                                  void *labels_out, int label_bytes, int32_t *bad_dev, void *stream)
{
    if (n < 0 || k < 1 || k > NNC_KMAX || !words_dev || !chunk_off_dev || !tables_dev || !bad_dev || (n > 0 && !labels_out) || (label_bytes != 1 && label_bytes != 2))
        return nnc_set_error_(NNC_EINVAL, "nnc_huffman_decode: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(bad_dev, 0, 4, s) != hipSuccess) return nnc_set_error_(NNC_EHIP, "nnc_huffman_decode: memset");
    const long long nchunks = (long long)nnc_codec_chunks(n);
    if (nchunks == 0) return NNC_OK;
    const uint32_t *fc = reinterpret_cast<const uint32_t *>(tables_dev), *cn = fc + (CODEC_MAXLEN + 1), *fi = cn + (CODEC_MAXLEN + 1);
    const uint16_t *syms = reinterpret_cast<const uint16_t *>(fi + (CODEC_MAXLEN + 1));
    const unsigned long long *co = reinterpret_cast<const unsigned long long *>(chunk_off_dev);
    const unsigned grid = (unsigned)((nchunks + 255) / 256);
    if (label_bytes == 1) hipLaunchKernelGGL((k_huff_decode<uint8_t>), dim3(grid), dim3(256), 0, s, words_dev, co, (long long)n, nchunks, fc, cn, fi, syms, (int)k, reinterpret_cast<uint8_t *>(labels_out), bad_dev);
    else hipLaunchKernelGGL((k_huff_decode<uint16_t>), dim3(grid), dim3(256), 0, s, words_dev, co, (long long)n, nchunks, fc, cn, fi, syms, (int)k, reinterpret_cast<uint16_t *>(labels_out), bad_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

// ------------------------------------------------------------------------------------------------------------------------
// Relative-index sparse form (Deep Compression section 3, the format the reference's report cites: papers/lat/report.tex:327,
// README.md:9): only the weights whose centroid index is NOT the zero cluster's are stored, each as (distance to the previous
// stored position, centroid index); the distance has `delta_bits` bits (stored value = distance - 1, distance in 1 .. D = 2^bits),
// and a gap longer than D takes filler entries (distance D, index = the zero cluster's) -- the paper's "padding zero".
// Distances restart at every chunk of NNC_CODEC_CHUNK positions (the position in front of a chunk counts as stored), so chunks
// encode and decode independently; entries_off[c] = number of entries in front of chunk c.  The two entry streams (distances,
// indices) are then Huffman coded by the functions above, as the paper does.
//
// One workgroup per chunk, a thread per four consecutive positions.  sp_scan_chunk: for each of a thread's positions the
// number of entries it emits (fillers + 1, or 0 for a zero-cluster position) and the distance its own entry carries.
template <typename LT>
__device__ __forceinline__ unsigned sp_scan_chunk(const LT *__restrict__ labels, long long n, long long base, int zero, int dbits, int *prev_s /*[4] LDS*/,
                                                  unsigned (&ent)[4], unsigned (&dist)[4], int (&sym)[4])
{
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const unsigned D = 1u << dbits;
    int last = -1; // last stored position (chunk-relative) among this thread's four, -1 if none
#pragma unroll
    for (int u = 0; u < 4; u++) {
        const long long g = base + tid * 4 + u;
        sym[u] = (g < n) ? (int)labels[g] : zero;
        if (sym[u] != zero) last = tid * 4 + u;
    }
    // previous stored position in front of this thread: inclusive max-scan over threads, shifted by one
    int inc = last;
    for (int o = 1; o < 64; o <<= 1) { const int t = __shfl_up(inc, o); if (lane >= o && t > inc) inc = t; }
    if (lane == 63) prev_s[wv] = inc;
    __syncthreads();
    int prev = __shfl_up(inc, 1);
    if (lane == 0) prev = -1;
    for (int w = 0; w < wv; w++) if (prev_s[w] > prev) prev = prev_s[w];
    unsigned total = 0;
#pragma unroll
    for (int u = 0; u < 4; u++) {
        ent[u] = 0; dist[u] = 0;
        if (sym[u] != zero) {
            const unsigned gap = (unsigned)(tid * 4 + u - prev); // >= 1
            const unsigned fill = (gap - 1) >> dbits;
            ent[u] = fill + 1;
            dist[u] = gap - fill * D; // 1 .. D
            prev = tid * 4 + u;
            total += ent[u];
        }
    }
    return total;
}

template <typename LT>
__global__ __launch_bounds__(256) void k_sparse_count(const LT *__restrict__ labels, long long n, int zero, int dbits, unsigned long long *__restrict__ chunk_entries)
{
    __shared__ int prev_s[4];
    __shared__ unsigned wsum[4];
    unsigned ent[4], dist[4];
    int sym[4];
    unsigned mine = sp_scan_chunk<LT>(labels, n, (long long)blockIdx.x * CODEC_CHUNK, zero, dbits, prev_s, ent, dist, sym);
    for (int off = 32; off >= 1; off >>= 1) mine += __shfl_xor(mine, off);
    if ((threadIdx.x & 63) == 0) wsum[threadIdx.x >> 6] = mine;
    __syncthreads();
    if (threadIdx.x == 0) chunk_entries[blockIdx.x] = (unsigned long long)wsum[0] + wsum[1] + wsum[2] + wsum[3];
}

template <typename LT>
__global__ __launch_bounds__(256) void k_sparse_emit(const LT *__restrict__ labels, long long n, int zero, int dbits, const unsigned long long *__restrict__ entries_off,
                                                     uint8_t *__restrict__ delta_out, LT *__restrict__ sym_out)
{
    __shared__ int prev_s[4];
    __shared__ unsigned wave_tot[4];
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    unsigned ent[4], dist[4];
    int sym[4];
    const unsigned mine = sp_scan_chunk<LT>(labels, n, (long long)blockIdx.x * CODEC_CHUNK, zero, dbits, prev_s, ent, dist, sym);
    unsigned inc = mine;
    for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
    if (lane == 63) wave_tot[wv] = inc;
    __syncthreads();
    unsigned long long pos = entries_off[blockIdx.x] + (inc - mine);
    for (int w = 0; w < wv; w++) pos += wave_tot[w];
    const uint8_t full = (uint8_t)((1u << dbits) - 1u);
#pragma unroll
    for (int u = 0; u < 4; u++) {
        if (!ent[u]) continue;
        for (unsigned f = 0; f + 1 < ent[u]; f++) { delta_out[pos] = full; sym_out[pos] = (LT)zero; pos++; } // padding zeros
        delta_out[pos] = (uint8_t)(dist[u] - 1u);
        sym_out[pos] = (LT)sym[u];
        pos++;
    }
}

// The inverse: a chunk's positions are filled with the zero cluster's index, then its entries are walked (running sum of the
// distances, 256 entries a round) and written.  *bad = 1 if an entry points outside its chunk.
template <typename LT>
__global__ __launch_bounds__(256) void k_sparse_expand(const uint8_t *__restrict__ delta, const LT *__restrict__ sym, const unsigned long long *__restrict__ entries_off,
                                                       long long n, int zero, LT *__restrict__ labels_out, int *__restrict__ bad)
{
    __shared__ unsigned wave_tot[4];
    __shared__ unsigned carry_s;
    const int tid = threadIdx.x, lane = tid & 63, wv = tid >> 6;
    const long long base = (long long)blockIdx.x * CODEC_CHUNK;
    const unsigned len = (unsigned)((base + CODEC_CHUNK < n ? base + CODEC_CHUNK : n) - base);
    for (unsigned i = tid; i < len; i += 256) labels_out[base + i] = (LT)zero;
    if (tid == 0) carry_s = 0;
    __syncthreads(); // the fill is visible to the whole workgroup before any entry is written over it
    const unsigned long long e0 = entries_off[blockIdx.x], e1 = entries_off[blockIdx.x + 1];
    for (unsigned long long e = e0; e < e1; e += 256) {
        const bool have = e + tid < e1;
        const unsigned d = have ? (unsigned)delta[e + tid] + 1u : 0u;
        unsigned inc = d;
        for (int o = 1; o < 64; o <<= 1) { const unsigned t = __shfl_up(inc, o); if (lane >= o) inc += t; }
        if (lane == 63) wave_tot[wv] = inc;
        __syncthreads();
        unsigned p = carry_s + inc;
        for (int w = 0; w < wv; w++) p += wave_tot[w];
        if (have) {
            if (p - 1u < len) labels_out[base + p - 1u] = sym[e + tid]; // position = (chunk start - 1) + running sum
            else *bad = 1;
        }
        __syncthreads();
        if (tid == 255) carry_s = p;
        __syncthreads();
    }
}

// entries_off_dev: (nchunks + 1) uint64; on return entries_off_dev[c] = entries in front of chunk c, [nchunks] = all entries.
extern "C" int nnc_sparse_entry_offsets(const void *labels, int label_bytes, int64_t n, int32_t zero_symbol, int32_t delta_bits, uint64_t *entries_off_dev, void *stream)
{
    if (n < 0 || !entries_off_dev || (n > 0 && !labels) || (label_bytes != 1 && label_bytes != 2) || delta_bits < 1 || delta_bits > 8 || zero_symbol < 0 || zero_symbol >= NNC_KMAX)
        return nnc_set_error_(NNC_EINVAL, "nnc_sparse_entry_offsets: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long nchunks = (long long)nnc_codec_chunks(n);
    unsigned long long *co = reinterpret_cast<unsigned long long *>(entries_off_dev);
    if (nchunks > 0) {
        if (label_bytes == 1) hipLaunchKernelGGL((k_sparse_count<uint8_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(labels), (long long)n, (int)zero_symbol, (int)delta_bits, co);
        else hipLaunchKernelGGL((k_sparse_count<uint16_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(labels), (long long)n, (int)zero_symbol, (int)delta_bits, co);
    }
    hipLaunchKernelGGL(k_scan_bits, dim3(1), dim3(1024), 0, s, co, nchunks);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

// delta_out_dev: uint8[entries] (distance - 1); sym_out_dev: entries indices of the labels' own width.
extern "C" int nnc_sparse_emit(const void *labels, int label_bytes, int64_t n, int32_t zero_symbol, int32_t delta_bits, const uint64_t *entries_off_dev,
                               uint8_t *delta_out_dev, void *sym_out_dev, void *stream)
{
    if (n < 0 || !entries_off_dev || (n > 0 && (!labels || !delta_out_dev || !sym_out_dev)) || (label_bytes != 1 && label_bytes != 2) || delta_bits < 1 || delta_bits > 8 ||
        zero_symbol < 0 || zero_symbol >= NNC_KMAX)
        return nnc_set_error_(NNC_EINVAL, "nnc_sparse_emit: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    const long long nchunks = (long long)nnc_codec_chunks(n);
    if (nchunks == 0) return NNC_OK;
    const unsigned long long *co = reinterpret_cast<const unsigned long long *>(entries_off_dev);
    if (label_bytes == 1) hipLaunchKernelGGL((k_sparse_emit<uint8_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint8_t *>(labels), (long long)n, (int)zero_symbol, (int)delta_bits, co, delta_out_dev, reinterpret_cast<uint8_t *>(sym_out_dev));
    else hipLaunchKernelGGL((k_sparse_emit<uint16_t>), dim3((unsigned)nchunks), dim3(256), 0, s, reinterpret_cast<const uint16_t *>(labels), (long long)n, (int)zero_symbol, (int)delta_bits, co, delta_out_dev, reinterpret_cast<uint16_t *>(sym_out_dev));
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}

extern "C" int nnc_sparse_expand(const uint8_t *delta_dev, const void *sym_dev, int label_bytes, const uint64_t *entries_off_dev, int64_t n, int32_t zero_symbol,
                                 void *labels_out, int32_t *bad_dev, void *stream)
{
    if (n < 0 || !entries_off_dev || !bad_dev || (n > 0 && !labels_out) || (label_bytes != 1 && label_bytes != 2) || zero_symbol < 0 || zero_symbol >= NNC_KMAX)
        return nnc_set_error_(NNC_EINVAL, "nnc_sparse_expand: bad argument");
    hipStream_t s = reinterpret_cast<hipStream_t>(stream);
    if (hipMemsetAsync(bad_dev, 0, 4, s) != hipSuccess) return nnc_set_error_(NNC_EHIP, "nnc_sparse_expand: memset");
    const long long nchunks = (long long)nnc_codec_chunks(n);
    if (nchunks == 0) return NNC_OK;
    const unsigned long long *co = reinterpret_cast<const unsigned long long *>(entries_off_dev);
    if (label_bytes == 1) hipLaunchKernelGGL((k_sparse_expand<uint8_t>), dim3((unsigned)nchunks), dim3(256), 0, s, delta_dev, reinterpret_cast<const uint8_t *>(sym_dev), co, (long long)n, (int)zero_symbol, reinterpret_cast<uint8_t *>(labels_out), bad_dev);
    else hipLaunchKernelGGL((k_sparse_expand<uint16_t>), dim3((unsigned)nchunks), dim3(256), 0, s, delta_dev, reinterpret_cast<const uint16_t *>(sym_dev), co, (long long)n, (int)zero_symbol, reinterpret_cast<uint16_t *>(labels_out), bad_dev);
    hipError_t e = hipGetLastError();
    if (e != hipSuccess) return nnc_set_error_(NNC_EHIP, hipGetErrorString(e));
    return NNC_OK;
}
