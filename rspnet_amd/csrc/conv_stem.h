// Internal interface of the direct "stem" convolution (conv_stem.hip), used by the rsp_conv3d_* entry points in
// conv_igemm.hip.  Not part of the C ABI.
#pragma once
#include "common.h"

// true when rsp_conv3d_{packed_fwd_elems,pack_fwd,stat_tiles,fwd} take the stem path for this descriptor
bool rsp_stem_applicable(const rsp_conv3d_desc* d);
int rsp_stem_tiles(const rsp_conv3d_desc* d);            // stat-partial tiles written by rsp_stem_fwd
size_t rsp_stem_packed_elems(const rsp_conv3d_desc* d);  // floats in the stem weight layout
int rsp_stem_pack(const rsp_conv3d_desc* d, const float* w_ref, float* w_packed, hipStream_t s);
int rsp_stem_fwd(const rsp_conv3d_desc* d, const float* x, const float* w_packed, const float* bias, float* y,
                 float* stat_partials, hipStream_t s);
const char* rsp_stem_kernel_name(const rsp_conv3d_desc* d);   // template instance rsp_stem_fwd launches
// the re-pack entry points record whether the filters behind a packed stem weight had three input channels (RGB): rsp_stem_fwd
// then skips the zero fourth channel's k-step
void rsp_stem_note_packed(const void* w_packed, bool three_channels);
void rsp_stem_forget_packed(const void* w_packed);
const char* rsp_wgrad_kernel_name(const rsp_conv3d_desc* d);
double rsp_wgrad_executed_fraction(const rsp_conv3d_desc* d);      // (conv_wgrad.hip) share of the row chunks its walk executes  // (conv_wgrad.hip) template instance rsp_conv3d_wgrad launches
