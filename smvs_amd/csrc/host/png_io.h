// PNG byte-image embeddings of an MVE view on top of zlib (csrc/host/png_io.cc).
#pragma once

#include <string>

#include "image.h"

namespace smvs_amd {

// 8 bits per sample; grey, grey + alpha, RGB, RGBA, palette (-> RGB);
// non-interlaced and Adam7.  Throws std::runtime_error otherwise.
ByteImage::Ptr load_png_u8(std::string const& path);
// width, height, channels (of the decoded image) without decoding it
bool png_header(std::string const& path, int* whc);
// 1 - 4 channels, filter type 0, one IDAT chunk
void save_png_u8(std::string const& path, ByteImage::ConstPtr image);

} // namespace smvs_amd
