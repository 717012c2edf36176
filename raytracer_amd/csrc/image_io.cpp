// Image output of the renderer: 8-bit conversion and PNG / PPM writers.  Replaces Image3f::write3fPpm
// and Image3f::save (src/Image3f.h:63-80,128-133,154-170 of the reference tree), the latter without
// OpenCV: the PNG is written directly (zlib deflate, filter type 0, 8-bit RGB).  The reference hands
// OpenCV a BGR buffer only because cv::imwrite expects that order; the file it writes is RGB.
#include "rtx.h"
#include "rtx_internal.h"

#include <zlib.h>

#include <cerrno>
#include <cstdio>
#include <cstring>
#include <string>
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

int write_png(const char* path, const uint8_t* rgb8, int w, int h) {
    std::vector<uint8_t> raw((size_t)h * (1 + 3 * (size_t)w));
    for (int y = 0; y < h; y++) {
        raw[(size_t)y * (1 + 3 * (size_t)w)] = 0;   // filter: none
        memcpy(&raw[(size_t)y * (1 + 3 * (size_t)w) + 1], rgb8 + (size_t)y * 3 * w, 3 * (size_t)w);
    }
    uLongf bound = compressBound((uLong)raw.size());
    std::vector<uint8_t> z(bound);
    if (compress2(z.data(), &bound, raw.data(), (uLong)raw.size(), 6) != Z_OK) {
        rtx_set_error("zlib compress2 failed");
        return RTX_E_IO;
    }
    std::vector<uint8_t> out;
    const uint8_t sig[8] = {0x89, 'P', 'N', 'G', 0x0d, 0x0a, 0x1a, 0x0a};
    out.insert(out.end(), sig, sig + 8);
    std::vector<uint8_t> ihdr;
    put_be32(ihdr, (uint32_t)w);
    put_be32(ihdr, (uint32_t)h);
    const uint8_t tail[5] = {8, 2, 0, 0, 0};   // 8 bits, colour type 2 (RGB), deflate, adaptive filtering, no interlace
    ihdr.insert(ihdr.end(), tail, tail + 5);
    put_chunk(out, "IHDR", ihdr.data(), ihdr.size());
    put_chunk(out, "IDAT", z.data(), bound);
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
