// msplat_kernels.hip.h -- hand-written CDNA4 (gfx950, wave64) kernels of the splat hot path, in eight parts.
//
// Replaces (not ports) the reference's GL pipeline:
//   shader/presort_compute.glsl + shader/multi_radixsort*.glsl  -> msplat_sort.hip.h      radix_* / ws_* (cull fused in pass 0)
//   shader/splat_vert.glsl + shader/splat_geom.glsl             -> msplat_project.hip.h   project_kernel
//   GL rasteriser + shader/splat_frag.glsl + ROP blend          -> msplat_binning.hip.h   bin1_* (+ radix_*<MODE_PAIR>)
//                                                                  msplat_composite.hip.h composite_kernel, composite_depth_kernel
//   GaussianCloud::ImportPly's per-vertex math, storage order   -> msplat_cloud.hip.h     ingest_kernel, morton / gather / cull boxes
//   shader/point_*.glsl                                         -> msplat_points.hip.h
//   two-pass frame with occlusion feedback (no reference part)  -> msplat_occlusion.hip.h
//   shared constants, FrameParams, cull_key, box_live           -> msplat_common.hip.h
//
// Design notes: everything is HBM/LDS/VALU work -- no MFMA anywhere.
// Compiled with -ffp-contract=off: an FMA happens only where __builtin_fmaf is written.
#pragma once

#include "msplat_common.hip.h"
#include "msplat_sort.hip.h"
#include "msplat_cloud.hip.h"
#include "msplat_project.hip.h"
#include "msplat_binning.hip.h"
#include "msplat_composite.hip.h"
#include "msplat_points.hip.h"
#include "msplat_occlusion.hip.h"
