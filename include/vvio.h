/* libvvio.so -- host-side frame codec of the MI355X VideoVanish build (plain C, no GPU): FFV1 (RFC 9043) version 3 encoder, version 0 / 1 / 3
 * decoder incl. non-key frames; 8-bit RGB (JPEG 2000 RCT) and planar YCbCr (decode), Golomb-Rice / range coder, slice CRCs -- the codec the reference writes with
 * cv2.VideoWriter(fourcc "FFV1") (reference tools.py:28-45) and reads back with cv2.VideoCapture (tools.py:4-25).
 * The Matroska container is written / parsed in Python (videovanish_amd/frameio.py); these entry points code single frames.
 * Source: videovanish_amd/csrc/vv_ffv1.c (built by csrc/build.sh with gcc); binding: videovanish_amd/frameio.py (ctypes).
 * Conventions: caller-owned buffers, no global state, thread safe, negative return = error. */
#ifndef VVIO_H
#define VVIO_H
#include <stdint.h>
#ifdef __cplusplus
extern "C" {
#endif

#define VVIO_ABI_VERSION 3
int vvio_abi_version(void);

/* FFV1 configuration record (the Matroska CodecPrivate payload after the BITMAPINFOHEADER; RFC 9043 section 4.2) for RGB24 frames cut
 * into num_v_slices horizontal slices.  Returns the record length, or -1 if cap is too small. */
int vvio_ffv1_config_record(int num_v_slices, uint8_t* out, int cap);

/* One RGB24 frame ([H][W][3], what tools.load_video_frames_from_path returns per frame) -> one FFV1 packet (key frame).
 * Returns the packet size, or -1 (buffer too small: cap >= 2 * W * H * 3 + 4096 is always enough).  Replaces VideoWriter.write
 * (reference tools.py:43). */
int vvio_ffv1_encode_frame(const uint8_t* rgb, int W, int H, int num_v_slices, uint8_t* out, int cap);

/* FFV1 packet + configuration record -> RGB24 (W*H*3 bytes).  Accepts what this encoder writes and the subset of FFV1 v3 streams with the
 * same colour model: 8-bit RGB (JPEG 2000 RCT), with or without an extra (alpha) plane (decoded and dropped); Golomb-Rice (coder_type 0)
 * or range-coded samples with the default (coder_type 1) or a custom (coder_type 2) state-transition table; any num_h_slices x num_v_slices
 * grid; up to 8 quantisation-table sets with 3 or 5 context inputs; CRC on or off.  Refused with an error code: YCbCr streams, > 8 bits per
 * sample, coded initial states, non-key frames (use the decoder object below), versions other than 3 (-2 / -26), and any header field outside
 * its range (the stream is untrusted input).  0 = ok, negative = error code.  Replaces VideoCapture.read (reference tools.py:17-21). */
int vvio_ffv1_decode_frame(const uint8_t* cfg, int cfglen, const uint8_t* data, int len, int W, int H, uint8_t* rgb);

/* ---- ABI 2: planar YCbCr streams (colorspace_type 0: what ffmpeg writes for yuv420p / yuv422p / yuv444p / gray FFV1) ----
 * vvio_ffv1_stream_info: info[6] = colorspace_type (0 YCbCr, 1 RGB), chroma_planes, log2_h_chroma_subsample, log2_v_chroma_subsample, extra_plane,
 * bits_per_raw_sample.  vvio_ffv1_decode_frame_yuv: Y [H][W], Cb / Cr [ceil(H >> vshift)][ceil(W >> hshift)] (gray: 128); Golomb-Rice or range-coded
 * samples, any slice grid, optional alpha plane (dropped).  vvio_ffv1_decode_frame refuses YCbCr streams with -9 and vice versa. */
int vvio_ffv1_stream_info(const uint8_t* cfg, int cfglen, int* info);
int vvio_ffv1_decode_frame_yuv(const uint8_t* cfg, int cfglen, const uint8_t* data, int len, int W, int H, uint8_t* y, uint8_t* cb, uint8_t* cr);
/* ---- ABI 3: the stateful decoder -- what a FILE needs (reference tools.py:4-25 reads whatever cv2.VideoWriter / ffmpeg wrote):
 *   * non-key frames (key-frame bit 0: no header, the adaptive context states continue from the previous frame, slice by slice; what an encoder
 *     with gop_size > 1 emits -- cv2.VideoWriter's default is 12 [UNVERIFIED-3P]);
 *   * FFV1 version 0 / 1 streams (cfglen = 0: no configuration record; parameters + one quantisation-table set in the header of every key frame,
 *     one slice, no slice header / footer / CRC -- libavcodec's own choice for frames up to 720 x 576 [UNVERIFIED-3P]).
 * vvio_ffv1_decoder_open: cfg / cfglen = the configuration record (version 3) or cfglen = 0 (version 0 / 1); *status = 0 or the error code.
 * vvio_ffv1_decoder_decode: the NEXT packet in stream order; returns 0 (RGB stream: rgb [H][W][3] filled), 1 (planar YCbCr stream: y / cb / cr
 * filled as vvio_ffv1_decode_frame_yuv does) or a negative error code (-20: a non-key frame without a preceding key frame).  Pass both buffer
 * kinds, the stream decides.  vvio_ffv1_decoder_info: info[7] = the six fields of vvio_ffv1_stream_info + the stream version (-30 before the
 * first key frame of a version 0 / 1 stream).  One decoder per stream; not shared between threads. */
void* vvio_ffv1_decoder_open(const uint8_t* cfg, int cfglen, int* status);
int vvio_ffv1_decoder_info(void* dec, int* info);
int vvio_ffv1_decoder_decode(void* dec, const uint8_t* data, int len, int W, int H, uint8_t* rgb, uint8_t* y, uint8_t* cb, uint8_t* cr);
void vvio_ffv1_decoder_close(void* dec);

/* YCbCr -> RGB24 on the host, the same integer arithmetic as the GPU kernel vv_ycbcr_to_rgb (include/vvhip.h): BT.601, limited or full range,
 * MPEG-2 4:2:0 chroma siting, bilinear chroma.  Used when no GPU is visible (frame I/O is host work; the hot path has no CPU fallback). */
int vvio_ycbcr_to_rgb(const uint8_t* y, const uint8_t* cb, const uint8_t* cr, int W, int H, int hshift, int vshift, int full_range, uint8_t* rgb);

#ifdef __cplusplus
}
#endif
#endif
