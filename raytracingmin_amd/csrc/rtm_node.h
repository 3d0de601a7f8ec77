// rtm_node.h — single-process multi-GPU render + RCCL gather used by rtm_cli (rtm_node.cpp).
#pragma once
#include <string>

#include "../../include/rtm.h"

// Renders the full frame as `n_devices` parts of interleaved 8-row bands (device r renders bands r,
// r + n_devices, ...) and gathers the 8-bit image on device 0 with one grouped ncclSend/ncclRecv
// exchange; virtual_strips > 0 instead renders that many parts one after another on options->device
// (same partition and assembly, no RCCL).
int rtm_node_render_u8(const rtm_settings* st, const rtm_sphere* spheres, size_t n, const rtm_options* base,
                       int n_devices, int virtual_strips, uint8_t* out_u8_host, rtm_stats* total,
                       std::string& err);
