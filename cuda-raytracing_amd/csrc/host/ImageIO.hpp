// ImageIO.hpp -- image files and the frame-display step without OpenCV.
//
// The reference decodes textures with cv::imread (Material.hpp:29-43) and ends every frame with
// display_image() (kernel.cu:30-43): download, "FPS: ..." overlay with cv::putText, cv::imwrite("out.png").
// This image has no OpenCV, so the decoders and the overlay are written here:
//   * PNG reader: zlib inflate (stored / fixed / dynamic Huffman blocks); grey, grey+alpha, RGB, RGBA and palette images
//     of every legal depth, plain or Adam7-interlaced; lossless, so its pixels are pinned by any other PNG decoder
//     (tests: Pillow).
//   * JPEG reader (sequential and progressive): see read_jpeg_bgr.
//   * text overlay with a built-in 5x7 font (cv::putText draws Hershey strokes; glyph shapes are NOT reproduced,
//     only position, colour and content of the overlay).
// All pixel buffers are B,G,R like cv::Mat.
#pragma once
#include <cstddef>
#include <cstdint>
#include <string>
#include <vector>
#include <hip/hip_vector_types.h>

#include "transforms.hpp"

using namespace transforms;      // as the reference's sources do (kernel.cu uses lre unqualified)

// ---- decoders: file -> tight B,G,R rows.  Return false (and leave the outputs alone) on anything malformed or unsupported.
bool read_png_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error = nullptr);
bool read_ppm_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error = nullptr);
bool read_jpeg_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error = nullptr);
// by file signature (PNG / JPEG / P6)
bool read_image_bgr(const std::string& path, std::vector<uint8_t>& bgr, int& width, int& height, std::string* error = nullptr);

// zlib stream -> bytes (used by the PNG reader; exposed for tests).  max_out: fail once the output would exceed this
// many bytes (deflate expands up to ~1000-fold: a reader passes the size its header announces)
bool zlib_inflate(const uint8_t* src, size_t n, std::vector<uint8_t>& out, std::string* error = nullptr,
                  size_t max_out = (size_t)1 << 30);

// ---- overlay: `text` with its baseline-left corner at (x, y) like cv::putText's `org`, glyph cell 5x7 scaled by `scale`
void overlay_text_bgr(uint8_t* bgr, int width, int height, size_t pitch, const std::string& text, int x, int y, int scale,
                      uint8_t b, uint8_t g, uint8_t r);

// ---- the reference's interaction state and handlers (kernel.cu:19-28, :51-139) -------------------------------
struct MouseParams {
    int last_x;
    int last_y;
    bool has_last = false;
    bool is_down = false;
    lre* pose;
};
// cv::MouseEventTypes values used by on_mouse
enum { RT_EVENT_MOUSEMOVE = 0, RT_EVENT_LBUTTONDOWN = 1, RT_EVENT_LBUTTONUP = 4 };
// kernel.cu:112-139: left-button drag turns the camera, yaw += dx * 0.001, pitch += dy * -0.001 (double arithmetic)
void on_mouse(int event, int x, int y, int flags, void* param);
// kernel.cu:51-103 (commented out in the snapshot): 'w' 's' 'a' 'd' move the camera 0.1 along its own y / x axes
// (new position = apply_lre(invert_lre(pose), step)); returns false for 'q' (the reference exits), true otherwise.
bool on_key(int key, MouseParams& mouse_state);

// kernel.cu:30-43: download the frame, overlay "FPS: <std::to_string(fps)>" in green at (10, 30), write out.png.
// `path` defaults to the reference's file name; `stream` = the stream the frame was rendered on (Camera::stream), the
// download is ordered behind it.  Returns an rt error code.
int display_image(const uchar3* d_img, int width, int height, size_t pitch, double fps, MouseParams& mouse_state,
                  const char* path = "out.png", void* stream = nullptr);
