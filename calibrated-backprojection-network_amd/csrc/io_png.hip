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

int png_decode_impl(const unsigned char* file, size_t file_bytes, void* pixels, size_t pixels_bytes) {
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
    std::vector<unsigned char> raw(raw_bytes);
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (inflateInit(&zs) != Z_OK) return KBN_ERR_LAUNCH;
    zs.next_out = raw.data();
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
    std::vector<unsigned char> zero(stride, 0);
    unsigned char* out8 = static_cast<unsigned char*>(pixels);
    for (int y = 0; y < h.height; ++y) {
        unsigned char* cur = raw.data() + (size_t)y * (stride + 1) + 1;
        const unsigned char* up = y ? cur - (stride + 1) : zero.data();
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
        return png_decode_impl(file, file_bytes, pixels, pixels_bytes);
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
            for (int i = next.fetch_add(1); i < n; i = next.fetch_add(1))
                st[i] = kbn_png_decode(files[i], file_bytes[i], pixels[i], pixels_bytes[i]);
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
