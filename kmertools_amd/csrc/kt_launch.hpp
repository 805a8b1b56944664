// kt_launch.hpp - host-side launch helpers shared by the translation units that use the
// segment front-end (kt_ctr.hip, kt_cov.hip).  Defined in kt_ctr.hip.
#pragma once
#include "kt_internal.hpp"
#include "kt_segment.hpp"
#include "kt_table.hpp"

namespace ktl {

// min(work_items, n_cu * per_cu) workgroups, at least 1
uint32_t grid_for(const kt_ctx *ctx, uint64_t work_items, uint32_t per_cu);
// SegArgs for a device-resident CSR batch (launches seg_index_kernel into ctx scratch s_aux0)
int make_seg_args(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                  uint64_t total_bases, int k, ktseg::SegArgs *out);
// offsets[n_reads]; a device read-back for KT_MEM_DEVICE
int total_bases_of(kt_ctx *ctx, const uint64_t *offsets, uint64_t n_reads, int mem, uint64_t *total);
// copies a host CSR batch into ctx scratch (s_bases, s_offsets)
int stage_batch(kt_ctx *ctx, const uint8_t *bases, const uint64_t *offsets, uint64_t n_reads,
                const uint8_t **d_bases, const uint64_t **d_offsets);
// the table's geometry as the device functions take it (kt_table.hpp)
inline kttab::Geom geom_of(const kt_ctr *ctr) { return kttab::Geom{ctr->cap, ctr->shift, ctr->m8, ctr->kbits}; }
// makes the table readable: performs a deferred clear, reports KT_ERR_FULL if it overflowed
int table_ready(kt_ctr *ctr);

}  // namespace ktl
