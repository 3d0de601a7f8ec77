// rtm_node.h — single-process multi-GPU render + RCCL gather used by rtm_cli (rtm_node.cpp).
#pragma once
#include <string>

#include "../../include/rtm.h"

// Renders the full frame as `n_devices` parts of interleaved 8-row bands (device r renders bands r,
// r + n_devices, ...) and gathers the float3 image and its 8-bit view on device 0 with one grouped
// ncclSend/ncclRecv exchange (taken for n_devices > 1, and for n_devices == 1 when force_rccl is set:
// communicator init + a grouped self send/recv); virtual_strips > 0 instead renders that many parts one
// after another on options->device (same partition and assembly, no RCCL).  Either output may be null.
int rtm_node_render(const rtm_settings* st, const rtm_object* objects, size_t n, const rtm_options* base,
                    int n_devices, int virtual_strips, int force_rccl, float* out_f32_host, uint8_t* out_u8_host,
                    rtm_stats* total, std::string& err);
