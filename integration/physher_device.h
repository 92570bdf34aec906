/*
 * physher_device.h -- what physher gains from seam A: a device backend for
 * SingleTreeLikelihood.  TEST INFRASTRUCTURE on this side of the boundary: the file that implements it
 * (physher_device.c) is compiled against the reference's own headers (-I$REF/src) and linked with the
 * compiled reference (oracle/_ref/libphyc_ref.so) and the product (physher_amd/libphysher_amd.so); nothing
 * under physher_amd/ or in bench.py's timed region links it.
 */
#ifndef PHYSHER_DEVICE_H
#define PHYSHER_DEVICE_H

#include "phyc/treelikelihood.h"

/* Move the hot path of `tlk` (post-order pass, root integration, pre-order pass, branch / site / substitution
 * gradients, single-branch evaluations, store / restore) to `device_count` GPUs (device_ids NULL: 0 .. device_count-1;
 * device_count <= 0: one engine on the current device).  `model` is the TreeLikelihood Model that owns tlk (NULL for a
 * bare SingleTreeLikelihood: store / restore and d2logP are then left on the CPU path).  Returns 0, or a negative
 * PHYAMD_E* code with the message on stderr -- the object then stays on the CPU kernels. */
int SingleTreeLikelihood_enable_device(SingleTreeLikelihood *tlk, Model *model, int device_count, const int *device_ids);
/* back to the CPU kernels (function pointers restored, engine destroyed) */
void SingleTreeLikelihood_disable_device(SingleTreeLikelihood *tlk);
/* 1 if tlk runs on the device */
int SingleTreeLikelihood_on_device(const SingleTreeLikelihood *tlk);
/* 1 if the device engine has switched to rescaled evaluations (the CPU-side flag tlk->scale is only the caller's request) */
int SingleTreeLikelihood_device_is_rescaling(const SingleTreeLikelihood *tlk);

#endif
