// Image buffer and 32-bit BMP writer: same API and file layout as the reference's
// image.h / image.cpp (54-byte header, top-down rows via negative height, pixels
// stored as little-endian 0xAARRGGBB words, i.e. B,G,R,A bytes).
#pragma once

#include <cstdint>

namespace rth {

struct Image {
    int32_t width = 0;
    int32_t height = 0;
    uint32_t* pixelData = nullptr;
};

Image CreateImage(int32_t width, int32_t height);
// Returns false if the file cannot be written (the reference ignores the failure, image.cpp:33-34).
bool WriteImageFile(const Image* image, const char* filename);
void FreeImage(Image* image);

}  // namespace rth
