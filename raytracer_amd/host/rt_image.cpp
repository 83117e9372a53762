#include "rt_image.h"

#include <cstdio>
#include <cstdlib>
#include <cstring>

namespace rth {

Image CreateImage(int32_t width, int32_t height) {
    Image image;
    image.width = width;
    image.height = height;
    image.pixelData = static_cast<uint32_t*>(std::calloc((size_t)width * (size_t)height, sizeof(uint32_t)));
    return image;
}

void FreeImage(Image* image) {
    std::free(image->pixelData);
    image->pixelData = nullptr;
}

namespace {
// Little-endian field writer for the two BMP headers (image.h:8-32 declares them as packed structs).
struct ByteWriter {
    unsigned char* p;
    void u16(uint16_t v) { p[0] = (unsigned char)v; p[1] = (unsigned char)(v >> 8); p += 2; }
    void u32(uint32_t v) { for (int i = 0; i < 4; ++i) p[i] = (unsigned char)(v >> (8 * i)); p += 4; }
};
}  // namespace

bool WriteImageFile(const Image* image, const char* filename) {
    const uint32_t pixel_bytes = (uint32_t)image->width * (uint32_t)image->height * 4u;
    unsigned char header[54];
    ByteWriter w{header};
    w.u16(0x4D42);                 // "BM"
    w.u32(54u + pixel_bytes);      // file size
    w.u16(0); w.u16(0);            // reserved
    w.u32(54);                     // offset of the pixel data
    w.u32(40);                     // info header size
    w.u32((uint32_t)image->width);
    w.u32((uint32_t)(-image->height));   // negative: rows are stored top-down
    w.u16(1);                      // planes
    w.u16(32);                     // bits per pixel
    w.u32(0);                      // no compression
    w.u32(0);                      // size of bitmap (may be 0 for uncompressed)
    w.u32(0); w.u32(0);            // resolution
    w.u32(0); w.u32(0);            // palette
    FILE* f = std::fopen(filename, "wb");
    if (!f) return false;
    bool ok = std::fwrite(header, sizeof header, 1, f) == 1 &&
              (pixel_bytes == 0 || std::fwrite(image->pixelData, pixel_bytes, 1, f) == 1);
    return std::fclose(f) == 0 && ok;
}

}  // namespace rth
