// b9_launch.h -- host-callable launch wrappers of the kernels in b9_kernels.hip.
#pragma once
#include "b9_device.h"

// where k_finalize writes the NEXT step's proposal and isochrone(s) (fused device-resident sampler)
struct B9Next { double *params; IsoHdr *hdr; double *iso; };

hipError_t b9k_derive_iso(const DevPack &pk, double *d_params, int n_walkers, int n_pops,
                          IsoHdr *hdr, double *iso_data, long long iso_stride, int mass_cap,
                          const McmcDev &mc, hipStream_t stream);

size_t b9k_star_like_lds_bytes(int n_pops, int mass_cap, int wb);

hipError_t b9k_star_like(const DevPack &pk, const DevStars &st, const IsoHdr *hdr,
                         const double *iso_data, long long iso_stride, int mass_cap,
                         const double *d_params, int n_walkers, int n_pops, int wb,
                         double *partial, double *perstar, int tiles_per_block, int n_groups,
                         hipStream_t stream);

hipError_t b9k_finalize(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                        long long iso_stride, int mass_cap, const double *partial, int n_partial, int n_pops,
                        const double *d_params, const DevPriors &pr, int n_walkers, double *d_logpost,
                        double *perstar, const McmcDev &mc, bool marg, const B9Next &nx, hipStream_t stream);

hipError_t b9k_star_marg(const DevPack &pk, const DevStars &st, const IsoHdr *hdr, const double *iso_data,
                         long long iso_stride, int mass_cap, const double *d_params, int n_walkers, int n_pops,
                         double *vals, double *perstar, int K, int Q, hipStream_t stream);
