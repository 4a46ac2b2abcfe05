/* jsplayer_amd_lab.h — MEASUREMENT helpers exported by libjsplayer_amd.so beside the drop-in boundary (include/jsplayer_amd.h).
 * Nothing here has a counterpart in the reference and no caller of the codec needs it: bench.py uses the two entry points to print
 * what THIS box's memory and bus deliver next to what the decode kernels reach. */
#ifndef JSPLAYER_AMD_LAB_H
#define JSPLAYER_AMD_LAB_H

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* The store rate this GPU reaches when asked for nothing else — `reps` launches
 * that fill `nbytes` of `device` (16-byte aligned) with one 16-byte store per lane, workgroups in address order, timed with HIP
 * events on `hip_stream`; best of three passes, GB/s.  bench.py reports it next to the 8 TB/s the roofline is priced against
 * (the boxes of one pool differ by a fifth in what their memory delivers). */
int jsp_measure_fill(int32_t* device, size_t nbytes, int reps, double* gbytes_per_s, void* hip_stream);
/* ... and what the BUS delivers (no reference counterpart): `copies` pinned host-to-device copies of `bytes_per_copy` on each of `nstreams`
 * (1..16) HIP streams of the device side by side, wall clock, best of three passes, GB/s — the ceiling of every end-to-end rate, where
 * the compressed bytes are all that crosses (bench.py: e2e.h2d_ceiling_GBs). */
int jsp_measure_h2d(int device_id, size_t bytes_per_copy, int nstreams, int copies, double* gbytes_per_s);

#ifdef __cplusplus
}
#endif
#endif /* JSPLAYER_AMD_LAB_H */
