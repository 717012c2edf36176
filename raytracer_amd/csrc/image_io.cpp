// Image output of the renderer: 8-bit conversion and PNG / PPM writers.  Replaces Image3f::write3fPpm
// and Image3f::save (src/Image3f.h:63-80,128-133,154-170 of the reference tree), the latter without
// OpenCV: the PNG is written directly (zlib deflate, bands of rows compressed side by side, filter type 0, 8-bit RGB).  The reference hands
// OpenCV a BGR buffer only because cv::imwrite expects that order; the file it writes is RGB.
#include "rtx.h"
#include "rtx_internal.h"

#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <algorithm>
#include <atomic>
#include <string>
#include <thread>
#include <vector>

static thread_local char g_err[512] = "";

void rtx_set_error(const char* fmt, ...) {
    va_list ap;
    va_start(ap, fmt);
    vsnprintf(g_err, sizeof g_err, fmt, ap);
    va_end(ap);
}

namespace {

// (unsigned char)(clamp(c) * 255) with clamp = max(min(c, 1), 0) in select form (src/Image3f.h:15-17,72-74);
// a NaN survives the clamp and converts to 0 here (the reference's conversion of NaN is undefined).
inline uint8_t to_u8(float c) {
    float hi = (1.0f < c) ? 1.0f : c;
    float v = (hi < 0.0f) ? 0.0f : hi;
    float s = v * 255;
    if (!(s == s)) return 0;
    return (uint8_t)s;
}

void put_be32(std::vector<uint8_t>& v, uint32_t x) {
    v.push_back((uint8_t)(x >> 24)); v.push_back((uint8_t)(x >> 16)); v.push_back((uint8_t)(x >> 8)); v.push_back((uint8_t)x);
}
void put_chunk(std::vector<uint8_t>& out, const char type[4], const uint8_t* data, size_t n) {
    put_be32(out, (uint32_t)n);
    size_t at = out.size();
    out.insert(out.end(), type, type + 4);
    if (n) out.insert(out.end(), data, data + n);
    uint32_t crc = (uint32_t)crc32(0L, out.data() + at, (uInt)(n + 4));
    put_be32(out, crc);
}

// One band of rows as a raw deflate stream of its own (no zlib header or trailer).  Every band but the last ends
// with a full flush, i.e. on a byte boundary and without the final-block flag, so that the bands' streams laid end to
// end are one valid deflate stream: the bands can be compressed side by side.
struct PngBand {
    int y0 = 0, y1 = 0;
    std::vector<uint8_t> z;
    uLong adler = 1, raw_bytes = 0;
    bool ok = false;
};
void deflate_band(PngBand* band, const uint8_t* rgb8, int w, bool last) {
    const size_t row = 1 + 3 * (size_t)w;
    std::vector<uint8_t> raw((size_t)(band->y1 - band->y0) * row);
    for (int y = band->y0; y < band->y1; y++) {
        uint8_t* dst = &raw[(size_t)(y - band->y0) * row];
        dst[0] = 0;   // filter: none
        memcpy(dst + 1, rgb8 + (size_t)y * 3 * w, 3 * (size_t)w);
    }
    band->raw_bytes = (uLong)raw.size();
    band->adler = adler32(adler32(0L, Z_NULL, 0), raw.data(), (uInt)raw.size());
    z_stream zs;
    memset(&zs, 0, sizeof zs);
    if (deflateInit2(&zs, 6, Z_DEFLATED, -15, 8, Z_DEFAULT_STRATEGY) != Z_OK) return;
    band->z.resize(deflateBound(&zs, (uLong)raw.size()) + 64);
    zs.next_in = raw.data();
    zs.avail_in = (uInt)raw.size();
    zs.next_out = band->z.data();
    zs.avail_out = (uInt)band->z.size();
    const int rc = deflate(&zs, last ? Z_FINISH : Z_FULL_FLUSH);
    band->ok = (last ? rc == Z_STREAM_END : rc == Z_OK) && zs.avail_in == 0 && zs.avail_out > 0;   // (room left: the flush is complete)
    band->z.resize(zs.total_out);
    deflateEnd(&zs);
}

int write_png(const char* path, const uint8_t* rgb8, int w, int h) {
    // bands of about half a megabyte of pixels (a band's stream cannot refer back into the band before it, which costs
    // compression on small bands), handed to at most sixteen threads one at a time: the rows of an image differ a lot
    // in how long they take to compress (the flat rows above and below an object take almost no time)
    const size_t row = 1 + 3 * (size_t)w;
    const int band_rows = (int)std::min<size_t>(std::max<size_t>(((size_t)1 << 19) / row, 1), (size_t)h);
    const int n_bands = (h + band_rows - 1) / band_rows;
    // (one band's input must fit zlib's 32-bit counts)
    if ((size_t)band_rows * row > 0x7fffffffu) { rtx_set_error("image too wide for the PNG writer"); return RTX_E_LIMIT; }
    std::vector<PngBand> bands((size_t)n_bands);
    for (int b = 0; b < n_bands; b++) {
        bands[b].y0 = b * band_rows;
        bands[b].y1 = std::min(h, (b + 1) * band_rows);
    }
    std::atomic<int> next{0};
    const auto work = [&]() {
        for (int b = next.fetch_add(1); b < n_bands; b = next.fetch_add(1)) deflate_band(&bands[b], rgb8, w, b == n_bands - 1);
    };
    const int n_threads = (int)std::min<size_t>({(size_t)std::max(1u, std::thread::hardware_concurrency()), (size_t)16, (size_t)n_bands});
    std::vector<std::thread> workers;
    for (int t = 1; t < n_threads; t++) workers.emplace_back(work);
    work();
    for (std::thread& t : workers) t.join();
    uLong adler = adler32(0L, Z_NULL, 0);
    size_t z_bytes = 2 + 4;
    for (const PngBand& band : bands) {
        if (!band.ok) { rtx_set_error("zlib deflate failed"); return RTX_E_IO; }
        adler = adler32_combine(adler, band.adler, (z_off_t)band.raw_bytes);
        z_bytes += band.z.size();
    }
    std::vector<uint8_t> z;
    z.reserve(z_bytes);
    z.push_back(0x78); z.push_back(0x9c);                      // zlib header: deflate, 32 KB window, default level
    for (const PngBand& band : bands) z.insert(z.end(), band.z.begin(), band.z.end());
    put_be32(z, (uint32_t)adler);
    std::vector<uint8_t> out;
    out.reserve(z.size() + 64);
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};   // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), z.size());
    put_chunk(out, "IEND", nullptr, 0);
    FILE* f = fopen(path, "wb");
    if (!f) { rtx_set_error("cannot write '%s': %s", path, strerror(errno)); return RTX_E_IO; }
    size_t wr = fwrite(out.data(), 1, out.size(), f);
    fclose(f);
    if (wr != out.size()) { rtx_set_error("short write to '%s'", path); return RTX_E_IO; }
    return RTX_OK;
}

int write_ppm(const char* path, const uint8_t* rgb8, int w, int h) {
    FILE* f = fopen(path, "wb");
    if (!f) { rtx_set_error("cannot write '%s': %s", path, strerror(errno)); return RTX_E_IO; }
    fprintf(f, "P6\n%d %d\n255\n", w, h);   // header layout of Image3f::write3fPpm
    size_t n = (size_t)w * h * 3;
    size_t wr = fwrite(rgb8, 1, n, f);
    fclose(f);
    if (wr != n) { rtx_set_error("short write to '%s'", path); return RTX_E_IO; }
    return RTX_OK;
}

}  // namespace

extern "C" {

const char* rtx_last_error(void) { return g_err; }

const char* rtx_strerror(int code) {
    switch (code) {
        case RTX_OK: return "ok";
        case RTX_E_INVALID: return "invalid argument";
        case RTX_E_IO: return "i/o error";
        case RTX_E_DEVICE: return "HIP device error";
        case RTX_E_STATE: return "call order violated";
        case RTX_E_NOMEM: return "out of memory";
        case RTX_E_LIMIT: return "size beyond a documented maximum";
        default: return "unknown error code";
    }
}

int rtx_quantize(const float* rgb, int64_t n, uint8_t* out) {
    if ((!rgb || !out) && n > 0) { rtx_set_error("rtx_quantize: null buffer"); return RTX_E_INVALID; }
    if (n < 0) { rtx_set_error("rtx_quantize: negative size"); return RTX_E_INVALID; }
    for (int64_t i = 0; i < n; i++) out[i] = to_u8(rgb[i]);
    return RTX_OK;
}

int rtx_save_image(const char* path, const float* rgb, int32_t w, int32_t h) {
    if (!path || !rgb || w <= 0 || h <= 0) { rtx_set_error("rtx_save_image: bad argument"); return RTX_E_INVALID; }
    std::vector<uint8_t> q;
    try {
        q.resize((size_t)w * h * 3);
    } catch (const std::bad_alloc&) {
        return RTX_E_NOMEM;
    }
    rtx_quantize(rgb, (int64_t)w * h * 3, q.data());
    return rtx_save_image_u8(path, q.data(), w, h);
}

int rtx_save_image_u8(const char* path, const uint8_t* rgb8, int32_t w, int32_t h) {
    if (!path || !rgb8 || w <= 0 || h <= 0) { rtx_set_error("rtx_save_image_u8: bad argument"); return RTX_E_INVALID; }
    std::string p(path);
    bool ppm = p.size() >= 4 && (p.compare(p.size() - 4, 4, ".ppm") == 0 || p.compare(p.size() - 4, 4, ".PPM") == 0);
    try {
        return ppm ? write_ppm(path, rgb8, w, h) : write_png(path, rgb8, w, h);
    } catch (const std::bad_alloc&) {
        return RTX_E_NOMEM;
    }
}

}  // extern "C"
