/*
 * yak_oracle.h -- CPU restatement of the Yak / kfusion TSDF hot path.
 *
 * TEST INFRASTRUCTURE ONLY.  Nothing under oracle/ is part of the product:
 * only tests/, __graft_entry__.smoke() and bench.py's cpu_baseline leg may
 * load it, and only as the checker / reported baseline.
 *
 * Parity status: the reference holds NO test, fixture or golden vector for
 * integrate / compute_dists / raycast / ICP, and its
 * implementation of them is CUDA-only, so it cannot be built here.  For those
 * functions this oracle is "PARITY UNPINNED": it is a line-by-line scalar
 * restatement of the cited CUDA sources, reviewed against them, and pinned
 * only where the reference does hold vectors:
 *   - voxel wire format (test/marching_cubes_tests.cpp:46-79),
 *   - half -> float decoding (cross-checked against the reference's vendored
 *     half.hpp compiled into oracle/_ref, all 65536 codes),
 *   - marching cubes dipyramid known answer (test/marching_cubes_tests.cpp:86-150).
 *
 * All file:line citations are relative to /root/reference/yak/.
 */
#ifndef YAK_ORACLE_H_
#define YAK_ORACLE_H_

#include <stddef.h>
#include <stdint.h>

#ifdef __cplusplus
extern "C" {
#endif

/* ---- half precision (cuda_fp16 __float2half[_rn] / __half2float) ---------------- */
uint16_t yko_float2half_rn(float f);
float yko_half2float(uint16_t h);
void yko_float2half_array(const float* in, uint16_t* out, size_t n);
void yko_half2float_array(const uint16_t* in, float* out, size_t n);

/* ---- voxel wire format: include/yak/kfusion/cuda/device.hpp:68-81 ---------------- */
uint32_t yko_pack_tsdf(float tsdf, int weight);
float yko_unpack_tsdf(uint32_t voxel, int* weight);

/* ---- src/cuda/imgproc.cu:259-283 (compute_dists_kernel + launcher) ---------------- */
void yko_compute_dists(const uint16_t* depth, size_t depth_pitch_bytes, uint16_t* dists, size_t dists_pitch_bytes,
                       int rows, int cols, float fx, float fy, float cx, float cy);

/* ---- src/cuda/tsdf_volume.cu:14-40 ---------------- */
void yko_clear_volume(uint32_t* volume, size_t n_voxels);

/*
 * ---- src/cuda/tsdf_volume.cu:51-140 (TsdfIntegrator + device::integrate) ----
 *
 * `slab` points at layer z0 of a volume whose full extent is dims[0..2]; the
 * layers [z0, z1) are updated.  The camera-space position is accumulated from
 * z = 0 exactly as the reference does (vc += zstep per step, tsdf_volume.cu:76)
 * so a slab gives bit-identical voxels to the same layers of a whole-volume run.
 *
 * R is row-major (precomp.hpp:15-30), intr = {fx, fy, cx, cy}.
 * stats (may be NULL): stats[0] += voxels visited, stats[1] += voxels updated.
 * nthreads <= 1: scalar, single thread.  > 1: OpenMP over y.
 */
void yko_integrate(uint32_t* slab, const int dims[3], int z0, int z1, const float voxel_size[3], float trunc_dist,
                   int max_weight, const uint16_t* dists, size_t dists_pitch_bytes, int rows, int cols,
                   const float R[9], const float t[3], const float intr[4], uint64_t stats[2], int nthreads);

/*
 * ---- src/cuda/tsdf_volume.cu:149-425 (intersect, interpolate, TsdfRaycaster, device::raycast) ----
 *
 * One ray per pixel from the camera through the volume (whole volume, dims[0..2]).
 * R, t = cam2vol (row-major R); Rinv = the inverse rotation the host forms
 * (tsdf_volume.cpp:117,136) -- an INPUT here, as it is for the device function.
 * intr = {fx, fy, cx, cy}; finv = 1/f is formed as in precomp.cpp:73-76.
 * points / normals: rows x cols float4, row pitch in bytes; depth (u16 mm) may be given instead of
 * points (the Depth overload, tsdf_volume.cu:220-285): pass the one wanted, NULL for the other.
 * Pixels without a hit hold quiet NaN x4 (points, normals) or 0 (depth).
 * rsqrt (temp_utils.hpp:103) -> 1.0f / sqrtf;  __fdividef -> '/';  __float2int_rn -> nearbyintf
 * (RNE); __float2int_rd -> floorf.  No FMA contraction (C semantics; nvcc's -fmad is unpinned).
 * fetch_tsdf (cu:210-217) has no bounds test in the reference; indices are clamped to the volume
 * here, which changes nothing wherever the reference's read is defined.
 * stats (may be NULL): stats[0] += nearest-voxel fetches, stats[1] += pixels with a hit.
 */
void yko_raycast(const uint32_t* volume, const int dims[3], const float voxel_size[3], float trunc_dist,
                 const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                 float raycast_step_factor, float gradient_delta_factor, int rows, int cols, float* points,
                 size_t points_pitch_bytes, float* normals, size_t normals_pitch_bytes, uint16_t* depth,
                 size_t depth_pitch_bytes, uint64_t stats[2], int nthreads);

/* The same march restricted to a z-slab's own layers [z_own0, z_own1) of the (whole) volume: a sample
 * pair is examined only if its first sample's nearest layer is one of them.  No counterpart in the
 * reference: it is the checker for yakhip_raycast_slab, whose per-slab results composite (smallest key
 * wins) into yko_raycast's image.  keys (int32 image, may be NULL) = 2 * index of the pair that ended
 * the march, INT32_MAX if none did. */
void yko_raycast_slab(const uint32_t* volume, const int dims[3], const float voxel_size[3], float trunc_dist,
                      const float R[9], const float t[3], const float Rinv[9], const float intr[4],
                      float raycast_step_factor, float gradient_delta_factor, int rows, int cols, float* points,
                      size_t points_pitch_bytes, float* normals, size_t normals_pitch_bytes, uint16_t* depth,
                      size_t depth_pitch_bytes, uint64_t stats[2], int nthreads, int z_own0, int z_own1, int32_t* keys,
                      size_t keys_pitch_bytes);

/* ---- pre-processing of the depth frame for ICP: src/cuda/imgproc.cu ------------------------- */

/* imgproc.cu:10-56 (bilateral_kernel + device::bilateralFilter).  sigma_depth in metres (scaled to mm
 * as :48 does).  __expf -> expf: the u16 result may differ from a GPU's by 1 where the quotient
 * sits next to a rounding boundary; tests allow exactly that.  __float2int_rn -> RNE, NaN -> 0. */
void yko_bilateral_filter(const uint16_t* src, size_t src_pitch_bytes, uint16_t* dst, size_t dst_pitch_bytes, int rows,
                          int cols, int kernel_size, float sigma_spatial, float sigma_depth);
/* imgproc.cu:66-85 */
void yko_truncate_depth(uint16_t* depth, size_t pitch_bytes, int rows, int cols, float max_dist_m);
/* imgproc.cu:94-136: dst is (src_rows/2) x (src_cols/2) (imgproc.cpp:15-19) */
void yko_depth_pyramid(const uint16_t* src, size_t src_pitch_bytes, int src_rows, int src_cols, uint16_t* dst,
                       size_t dst_pitch_bytes, float sigma_depth);
/* imgproc.cu:210-250: points/normals are rows x cols float4 */
void yko_compute_points_normals(const uint16_t* depth, size_t depth_pitch_bytes, int rows, int cols,
                                const float intr[4], float* points, size_t points_pitch_bytes, float* normals,
                                size_t normals_pitch_bytes);
/* imgproc.cu:348-394: outputs are (src_rows/2) x (src_cols/2) (imgproc.cpp:65-80) */
void yko_resize_points_normals(const float* points, size_t points_pitch_bytes, const float* normals,
                               size_t normals_pitch_bytes, int src_rows, int src_cols, float* points_out,
                               size_t points_out_pitch_bytes, float* normals_out, size_t normals_out_pitch_bytes);

/*
 * ---- projective ICP, one linearisation: src/cuda/proj_icp.cu:24-39, 74-102, 343-364 (Points path;
 *      USE_DEPTH is not defined, internal.hpp:8) + projective_icp.cpp:10-22 ----
 * R,t = the current estimate (curr -> prev camera), intr = level-0 intrinsics, level = pyramid level
 * (setLevelIntr divides by 1 << level).  sums[27] = sum over accepted pixels of row[i]*row[j], i <= j < 7,
 * i < 6, in the order of proj_icp.cu:122-335 (what StreamHelper::get unpacks, projective_icp.cpp:42-61),
 * accumulated in DOUBLE (the reference's float tree order is a schedule, not a specification).
 * counts[6] = pixels accepted, then rejected with code 40, 80, 120, 160, 200 (cu:78-101).
 * codes (may be NULL): rows x cols bytes, the per-pixel code.
 */
void yko_icp_accumulate(const float R[9], const float t[3], const float intr[4], int level, float dist_thres,
                        float angle_thres, const float* vcurr, size_t vcurr_pitch_bytes, const float* ncurr,
                        size_t ncurr_pitch_bytes, const float* vprev, size_t vprev_pitch_bytes, const float* nprev,
                        size_t nprev_pitch_bytes, int rows, int cols, double sums[27], uint64_t counts[6],
                        uint8_t* codes);

/*
 * ---- marching cubes: src/mc/marching_cubes.cpp:22-194 (getOffset, cubeOffsets, processCube, makeMesh) ----
 * PINNED by the reference's dipyramid known answer (test/marching_cubes_tests.cpp:86-150).
 * volume: X*Y*Z voxel words (TSDFContainer layout, tsdf_container.h:23-27, 45-49; weight UNSIGNED here).
 * tri_table: 256 x 16 ints, edge_connections: 12 x 2 ints -- the reference's own arrays when
 * oracle/_ref is built (src/mc/marching_cubes_tables.cpp), handed in by the caller.
 * Cubes are visited x outer, y, z inner (the serial order of makeMesh's loops, :164-170); each triangle
 * is 9 floats {v0.xyz, v1.xyz, v2.xyz}, each coordinate (float)(scale * (double)v) as makeMesh +
 * vtk2mesh store it (:175-183).  Writes at most `capacity` triangles; returns the number found.
 */
uint64_t yko_marching_cubes(const uint32_t* volume, const int dims[3], int min_weight, double scale,
                            const int* tri_table, const int* edge_connections, float* triangles, uint64_t capacity);
/* the same, the x slices in parallel (the reference's OpenMP loop, marching_cubes.cpp:163); same triangle order */
uint64_t yko_marching_cubes_mt(const uint32_t* volume, const int dims[3], int min_weight, double scale,
                               const int* tri_table, const int* edge_connections, float* triangles, uint64_t capacity,
                               int nthreads);

/* ---- cloud extraction: src/cuda/tsdf_volume.cu:464-800 (FullScan6, ExtractNormals) --------------
 * parity unpinned: the reference holds no vector for it (and its own output order is nondeterministic:
 * a global atomic counter, :616). */

/* tsdf_volume.cu:468-665 + tsdf_volume.cpp:147-168: every voxel (x, y, z < Z - 1) with weight != 0 and
 * tsdf != 1 gives one point per axis neighbour (x + 1, y + 1, z + 1, inside the volume) that also has
 * weight != 0, tsdf != 1 and the opposite strict sign: the linear zero crossing between the two voxel
 * CENTRES ((i + 0.5) * voxel_size), mapped by aff = (R, t) (the volume pose).  points: float4 {x, y, z, 0},
 * visiting order z, y, x, then dx, dy, dz.  Writes at most `capacity` points; returns the number found. */
uint64_t yko_extract_cloud(const uint32_t* volume, const int dims[3], const float voxel_size[3], const float R[9],
                           const float t[3], float* points, uint64_t capacity);

/* tsdf_volume.cu:672-763 + tsdf_volume.cpp:170-183: for every point, back in the volume frame
 * (Rinv * (p - t)), the normalised central-difference gradient of the trilinear tsdf, rotated by R;
 * NaN where the nearest voxel is within 2 of the volume's faces.  normals: float4 {x, y, z, 0}. */
void yko_extract_normals(const uint32_t* volume, const int dims[3], const float voxel_size[3], const float* points,
                         uint64_t n_points, const float R[9], const float t[3], const float Rinv[9],
                         float gradient_delta_factor, float* normals);

/* Debug display, imgproc.cu:400-561 (parity unpinned; the specular power is v^20 by multiplication where the
 * reference has the approximate __powf).  image: 4 bytes per pixel {b, g, r, 0}. */
void yko_render_image_points(const float* points, size_t points_pitch_bytes, const float* normals,
                             size_t normals_pitch_bytes, int rows, int cols, const float light_pose[3], uint8_t* image,
                             size_t image_pitch_bytes);
void yko_render_image_depth(const uint16_t* depth, size_t depth_pitch_bytes, const float* normals,
                            size_t normals_pitch_bytes, int rows, int cols, const float intr[4],
                            const float light_pose[3], uint8_t* image, size_t image_pitch_bytes);
void yko_render_tangent_colors(const float* normals, size_t normals_pitch_bytes, int rows, int cols, uint8_t* image,
                               size_t image_pitch_bytes);

int yko_max_threads(void);

#ifdef __cplusplus
}
#endif
#endif
