// encode.hpp -- restored RGB pixels -> the base64 text of a PNG file, on the device (encode.hip).
#pragma once
#include "common.hpp"

namespace ire {
size_t png_file_bytes(int h, int w);        // the PNG file: signature, IHDR, one IDAT of stored deflate blocks, IEND
size_t png_base64_chars(int h, int w);      // its base64 text ('=' padded, no terminator)
size_t png_scratch_bytes(int h, int w);     // device scratch per image (the file + checksum state); must be zero at first use
void encode_png_base64_launch(const unsigned char* d_rgb, int h, int w, unsigned char* d_scratch, unsigned char* d_chars, hipStream_t s);
}  // namespace ire
