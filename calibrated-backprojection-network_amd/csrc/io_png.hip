// io_png.hip -- host-side PNG reader of the input pipeline.
//
// The reference reads its inputs through PIL: data_utils.load_image (`Image.open(path).convert('RGB')`,
// reference src/data_utils.py:58-85) and data_utils.load_depth (16-bit PNG / 256, :123-152).  This is
// the same decode as plain C++ on top of zlib's inflate, so that a pool of host threads can fill
// pinned staging buffers without the Python GIL (ctypes releases it around the call):
// IHDR / PLTE / IDAT chunks, the five scanline filters, 8-bit gray / RGB / RGBA / palette and 16-bit
// gray, non-interlaced.  Pixels come out as H x W x C uint8, or H x W uint16 in host byte order;
// palette images are expanded to RGB (what convert('RGB') does).  No GPU work here.
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <zlib.h>

#include <atomic>
#include <thread>
#include <vector>

#include "../../include/kbnet_hip.h"

namespace {

struct PngHeader {
    int width, height, bit_depth, color_type, interlace;
    int channels_file;  // samples per pixel in the file
    int channels_out;   // samples per pixel we deliver (palette -> 3)
};

inline uint32_t be32(const unsigned char* p) { return ((uint32_t)p[0] << 24) | ((uint32_t)p[1] << 16) | ((uint32_t)p[2] << 8) | p[3]; }

const unsigned char kSignature[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};

int parse_header(const unsigned char* file, size_t n, PngHeader* h) {
    if (!file || n < 8 + 25 || memcmp(file, kSignature, 8) != 0) return KBN_ERR_INVALID_ARGUMENT;
    if (be32(file + 8) != 13 || memcmp(file + 12, "IHDR", 4) != 0) return KBN_ERR_INVALID_ARGUMENT;
    const unsigned char* d = file + 16;
    h->width = (int)be32(d);
    h->height = (int)be32(d + 4);
    h->bit_depth = d[8];
    h->color_type = d[9];
    h->interlace = d[12];
    if (h->width < 1 || h->height < 1 || h->width > 32767 || h->height > 32767 || d[10] != 0 || d[11] != 0)
        return KBN_ERR_INVALID_ARGUMENT;
    if (h->interlace != 0) return KBN_ERR_UNSUPPORTED;
    switch (h->color_type) {
        case 0: h->channels_file = 1; h->channels_out = 1; if (h->bit_depth != 8 && h->bit_depth != 16) return KBN_ERR_UNSUPPORTED; break;
        case 2: h->channels_file = 3; h->channels_out = 3; if (h->bit_depth != 8) return KBN_ERR_UNSUPPORTED; break;
        case 3: h->channels_file = 1; h->channels_out = 3; if (h->bit_depth != 8) return KBN_ERR_UNSUPPORTED; break;
        case 6: h->channels_file = 4; h->channels_out = 4; if (h->bit_depth != 8) return KBN_ERR_UNSUPPORTED; break;
        default: return KBN_ERR_UNSUPPORTED;  // gray + alpha
    }
    return KBN_OK;
}

inline int paeth(int a, int b, int c) {
    const int p = a + b - c;
    const int pa = abs(p - a), pb = abs(p - b), pc = abs(p - c);
    return (pa <= pb && pa <= pc) ? a : (pb <= pc ? b : c);
}

}  // namespace

extern "C" {

int kbn_png_info(const unsigned char* file, size_t file_bytes, int* width, int* height, int* channels,
                 int* bit_depth) {
    PngHeader h;
    const int rc = parse_header(file, file_bytes, &h);
    if (rc != KBN_OK) return rc;
    if (width) *width = h.width;
    if (height) *height = h.height;
    if (channels) *channels = h.channels_out;
    if (bit_depth) *bit_depth = h.bit_depth;
    return KBN_OK;
}

}  // extern "C"

namespace {

// `scratch`: the caller's reusable buffer for the inflated scanlines (a worker of kbn_png_decode_batch keeps one for all its files: a fresh
// 3.9 MB vector per KITTI triplet is an mmap + 940 page faults + an munmap per image, and concurrent mmap / munmap calls of one process
// serialise on its address-space lock -- what flattened the batch decoder at 16-32 threads)
int png_decode_impl(const unsigned char* file, size_t file_bytes, void* pixels, size_t pixels_bytes, std::vector<unsigned char>& scratch) {
    PngHeader h;
    int rc = parse_header(file, file_bytes, &h);
    if (rc != KBN_OK) return rc;
    if (!pixels) return KBN_ERR_INVALID_ARGUMENT;
    const size_t bps = (size_t)h.bit_depth / 8;                        // bytes per sample
    const size_t bpp = bps * h.channels_file;                          // bytes per pixel in the file
    const size_t stride = bpp * h.width;                               // bytes per scanline (without the filter byte)
    const size_t need = (size_t)h.height * h.width * h.channels_out * bps;
    if (pixels_bytes < need) return KBN_ERR_INVALID_ARGUMENT;

    // ---- walk the chunks: palette, concatenated IDAT payload ----
    unsigned char palette[256 * 3];
    int palette_entries = 0;
    std::vector<const unsigned char*> idat_ptr;
    std::vector<size_t> idat_len;
    size_t pos = 8;
    bool end = false;
    while (!end) {
        if (pos + 12 > file_bytes) return KBN_ERR_INVALID_ARGUMENT;    // truncated
        const size_t len = be32(file + pos);
        const unsigned char* type = file + pos + 4;
        const unsigned char* data = file + pos + 8;
        if (len > file_bytes || pos + 12 + len > file_bytes) return KBN_ERR_INVALID_ARGUMENT;
        if (memcmp(type, "IDAT", 4) == 0) {
            idat_ptr.push_back(data);
            idat_len.push_back(len);
        } else if (memcmp(type, "PLTE", 4) == 0) {
            if (len % 3 != 0 || len > 768) return KBN_ERR_INVALID_ARGUMENT;
            memcpy(palette, data, len);
            palette_entries = (int)(len / 3);
        } else if (memcmp(type, "IEND", 4) == 0) {
            end = true;
        }
        pos += 12 + len;
    }
    if (idat_ptr.empty() || (h.color_type == 3 && palette_entries == 0)) return KBN_ERR_INVALID_ARGUMENT;

    // ---- inflate into [height][1 + stride] ----
    const size_t raw_bytes = (size_t)h.height * (stride + 1);
    if (raw_bytes > 0xffffffffull) return KBN_ERR_UNSUPPORTED;         // zlib's avail_out is 32 bits wide
    if (scratch.size() < raw_bytes + stride) scratch.resize(raw_bytes + stride);   // grows only; [raw_bytes, + stride) = the zero scanline above row 0
    unsigned char* const raw = scratch.data();
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return KBN_ERR_LAUNCH;
    zs.next_out = raw;
    zs.avail_out = (uInt)raw_bytes;
    int zrc = Z_OK;
    for (size_t i = 0; i < idat_ptr.size() && zrc == Z_OK; ++i) {
        zs.next_in = const_cast<unsigned char*>(idat_ptr[i]);
        zs.avail_in = (uInt)idat_len[i];                               // a chunk length is a 32-bit field
        zrc = inflate(&zs, Z_NO_FLUSH);
    }
    const bool complete = (zs.avail_out == 0) && (zrc == Z_OK || zrc == Z_STREAM_END);
    inflateEnd(&zs);
    if (!complete) return KBN_ERR_INVALID_ARGUMENT;

    // ---- undo the scanline filters in place (PNG spec 9.2), then deliver ----
    unsigned char* const zero = raw + raw_bytes;
    memset(zero, 0, stride);
    unsigned char* out8 = static_cast<unsigned char*>(pixels);
    for (int y = 0; y < h.height; ++y) {
        unsigned char* cur = raw + (size_t)y * (stride + 1) + 1;
        const unsigned char* up = y ? cur - (stride + 1) : zero;
        const int filter = cur[-1];
        switch (filter) {
            case 0: break;
            case 1: for (size_t i = bpp; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + cur[i - bpp]); break;
            case 2: for (size_t i = 0; i < stride; ++i) cur[i] = (unsigned char)(cur[i] + up[i]); break;
            case 3:
                for (size_t i = 0; i < stride; ++i) {
                    const int left = i >= bpp ? cur[i - bpp] : 0;
                    cur[i] = (unsigned char)(cur[i] + ((left + up[i]) >> 1));
                }
                break;
            case 4:
                for (size_t i = 0; i < stride; ++i) {
                    const int left = i >= bpp ? cur[i - bpp] : 0, ul = i >= bpp ? up[i - bpp] : 0;
                    cur[i] = (unsigned char)(cur[i] + paeth(left, up[i], ul));
                }
                break;
            default: return KBN_ERR_INVALID_ARGUMENT;
        }
        if (h.color_type == 3) {            // palette -> RGB
            unsigned char* o = out8 + (size_t)y * h.width * 3;
            for (int x = 0; x < h.width; ++x) {
                const int idx = cur[x];
                if (idx >= palette_entries) return KBN_ERR_INVALID_ARGUMENT;
                o[3 * x] = palette[3 * idx]; o[3 * x + 1] = palette[3 * idx + 1]; o[3 * x + 2] = palette[3 * idx + 2];
            }
        } else if (h.bit_depth == 16) {     // big endian samples -> host order
            uint16_t* o = static_cast<uint16_t*>(pixels) + (size_t)y * h.width;
            for (int x = 0; x < h.width; ++x) o[x] = (uint16_t)((cur[2 * x] << 8) | cur[2 * x + 1]);
        } else {
            memcpy(out8 + (size_t)y * stride, cur, stride);
        }
    }
    return KBN_OK;
}

}  // namespace

extern "C" {

int kbn_png_decode(const unsigned char* file, size_t file_bytes, void* pixels, size_t pixels_bytes) {
    try {   // nothing may throw across the ABI (or out of a worker thread): allocation failures become a status
        std::vector<unsigned char> scratch;
        return png_decode_impl(file, file_bytes, pixels, pixels_bytes, scratch);
    } catch (...) {
        return KBN_ERR_WORKSPACE;
    }
}

// Decodes `n` files on `threads` host threads (work stealing over an atomic index; no Python, no GIL).
// status[i] receives the per-file code; the return value is the first failure or KBN_OK.
int kbn_png_decode_batch(const unsigned char* const* files, const size_t* file_bytes, void* const* pixels,
                         const size_t* pixels_bytes, int n, int threads, int* status) {
    if (!files || !file_bytes || !pixels || !pixels_bytes || n < 0) return KBN_ERR_INVALID_ARGUMENT;
    if (n == 0) return KBN_OK;
    if (threads < 1) threads = 1;
    if (threads > n) threads = n;
    try {
        std::vector<int> local(status ? 0 : n);
        int* st = status ? status : local.data();
        std::atomic<int> next(0);
        auto worker = [&]() noexcept {
            std::vector<unsigned char> scratch;   // one scanline buffer per worker, reused over its files
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1)) {
                try {
                    st[i] = png_decode_impl(files[i], file_bytes[i], pixels[i], pixels_bytes[i], scratch);
                } catch (...) {
                    st[i] = KBN_ERR_WORKSPACE;
                }
            }
        };
        std::vector<std::thread> pool;
        try {
            for (int t = 1; t < threads; ++t) pool.emplace_back(worker);
        } catch (...) {   // the host refused another thread: the ones that started and this one share the work
        }
        worker();
        for (auto& t : pool) t.join();
        for (int i = 0; i < n; ++i)
            if (st[i] != KBN_OK) return st[i];
        return KBN_OK;
    } catch (...) {
        return KBN_ERR_WORKSPACE;
    }
}

}  // extern "C"

// ------------------------------------------------------------------------------------------------------------------
// PNG writer of the output side: data_utils.save_depth (reference src/data_utils.py:154-167: np.uint32(z * 256) ->
// Image.fromarray(mode='I').save(path), which PIL writes as a 16-bit grayscale PNG, samples clipped at 65535) for the depth
// maps run_kbnet.py --save_outputs stores (src/kbnet.py:1018-1026).  The samples come from kbn_depth_to_u16_forward; this is
// the file: signature, IHDR (16-bit gray, non-interlaced), ONE IDAT (zlib deflate of the scanlines, filter 0, big-endian
// samples), IEND.  Readers recover the samples bit for bit (kbn_png_decode, PIL); the bytes of the file differ from PIL's
// (its encoder picks other filters) -- a PNG's content is its pixels.
namespace {

inline void put_be32(unsigned char* p, uint32_t v) { p[0] = v >> 24; p[1] = v >> 16; p[2] = v >> 8; p[3] = v; }

size_t put_chunk(unsigned char* out, const char type[4], const unsigned char* data, uint32_t n) {
    put_be32(out, n);
    memcpy(out + 4, type, 4);
    if (n) memcpy(out + 8, data, n);
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, out + 4, n + 4);
    put_be32(out + 8 + n, (uint32_t)crc);
    return (size_t)n + 12;
}

int png_encode_gray16_impl(const unsigned short* pixels, int width, int height, unsigned char* out, size_t out_capacity,
                           size_t* out_bytes, int level) {
    if (!pixels || !out || !out_bytes || width < 1 || height < 1 || level < -1 || level > 9) return KBN_ERR_INVALID_ARGUMENT;
    const size_t row = 1 + 2 * (size_t)width, raw = row * height;
    if (raw > 0xffffffffull) return KBN_ERR_UNSUPPORTED;
    std::vector<unsigned char> lines(raw);
    for (int y = 0; y < height; ++y) {
        unsigned char* l = lines.data() + row * y;
        l[0] = 0;   // filter type None
        const unsigned short* src = pixels + (size_t)width * y;
        for (int x = 0; x < width; ++x) { l[1 + 2 * x] = (unsigned char)(src[x] >> 8); l[2 + 2 * x] = (unsigned char)(src[x] & 0xff); }
    }
    uLongf zn = compressBound((uLong)raw);
    if (8 + 25 + 12 + (size_t)zn + 12 > out_capacity) return KBN_ERR_WORKSPACE;   // kbn_png_encode_gray16_bound says how much
    unsigned char* p = out;
    memcpy(p, kSignature, 8); p += 8;
    unsigned char ihdr[13];
    put_be32(ihdr, (uint32_t)width); put_be32(ihdr + 4, (uint32_t)height);
    ihdr[8] = 16; ihdr[9] = 0; ihdr[10] = 0; ihdr[11] = 0; ihdr[12] = 0;
    p += put_chunk(p, "IHDR", ihdr, 13);
    if (compress2(p + 8, &zn, lines.data(), (uLong)raw, level) != Z_OK) return KBN_ERR_LAUNCH;
    if (zn > 0x7fffffffUL) return KBN_ERR_UNSUPPORTED;
    put_be32(p, (uint32_t)zn);
    memcpy(p + 4, "IDAT", 4);
    uLong crc = crc32(0L, Z_NULL, 0);
    crc = crc32(crc, p + 4, (uInt)zn + 4);
    put_be32(p + 8 + zn, (uint32_t)crc);
    p += (size_t)zn + 12;
    p += put_chunk(p, "IEND", nullptr, 0);
    *out_bytes = (size_t)(p - out);
    return KBN_OK;
}

}  // namespace

extern "C" {

size_t kbn_png_encode_gray16_bound(int width, int height) {
    if (width < 1 || height < 1) return 0;
    const unsigned long long raw = (1ull + 2ull * (unsigned long long)width) * (unsigned long long)height;
    if (raw > 0xffffffffull) return 0;
    return 8 + 25 + 12 + (size_t)compressBound((uLong)raw) + 12;
}

int kbn_png_encode_gray16(const unsigned short* pixels, int width, int height, unsigned char* out, size_t out_capacity,
                          size_t* out_bytes, int level) {
    try {
        return png_encode_gray16_impl(pixels, width, height, out, out_capacity, out_bytes, level);
    } catch (...) {
        return KBN_ERR_WORKSPACE;
    }
}

}  // extern "C"
