// JPEG byte-image embeddings of an MVE view (csrc/host/jpeg_io.cc): sequential
// and progressive Huffman JPEG, 8 bits, grey or three components, decoded the
// way libjpeg decodes with its default parameters (what
// mve::image::load_jpg_file uses).
#pragma once

#include <string>

#include "image.h"

namespace smvs_amd {

// Throws std::runtime_error with the reason for files it does not decode
// (arithmetic-coded, lossless, 12-bit, CMYK) and for corrupt data.
ByteImage::Ptr load_jpeg_u8(std::string const& path);
// width, height, channels without decoding the image data
bool jpeg_header(std::string const& path, int* whc);

} // namespace smvs_amd
