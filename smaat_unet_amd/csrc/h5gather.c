/* Host-side helper of the input pipeline (smaat_unet_amd/h5lite.py, data.H5SampleSource): inflate the deflate-compressed
 * chunks that hold the wanted frames of ONE sample of the reference's HDF5 dataset ("images" [samples][T][H][W] float32,
 * chunked + gzip, /root/reference/create_datasets.py:33-40) and scatter them straight into the destination frames of a
 * pinned batch buffer -- what `imgs = np.array(self.dataset[index]); imgs[:num_input], imgs[-1]` of
 * /root/reference/utils/dataset_precip.py:69-75 amounts to, without an intermediate sample array.
 *
 * Plain C + zlib, called through ctypes (which releases the GIL): the gather threads of PrefetchLoader run it in parallel.
 * The chunk list (file offsets / sizes from the version-1 B-tree index) is prepared by the Python side.
 * Built as smaat_unet_amd/libsmaat_io.so by csrc/Makefile (gcc, host only: no GPU code in here).
 */
#define _GNU_SOURCE
#include <stdint.h>
#include <stdlib.h>
#include <string.h>
#include <unistd.h>
#include <zlib.h>

/* returns 0, or a negative code: -1 bad argument, -2 allocation, -3 short read, -4 inflate error / size mismatch */
int smaat_h5_gather(int fd, int n_chunks, const int64_t* off, const int32_t* nbytes, const int32_t* origin /* [n][3]: frame,
                    row, col of the chunk's first element inside the sample */,
                    const int32_t* cdim /* [3] chunk extent in frames, rows, cols */, const int32_t* ext /* [3] T, H, W */,
                    const int32_t* fmap /* [T]: destination frame of source frame f, or -1 = not wanted */, float* dst,
                    int64_t dst_frame_stride, int64_t dst_row_stride, float fill) {
    if (fd < 0 || n_chunks < 0 || !cdim || !ext || !fmap || !dst) return -1;
    const size_t cbytes = (size_t)cdim[0] * cdim[1] * cdim[2] * sizeof(float);
    int32_t maxz = 0;
    for (int i = 0; i < n_chunks; ++i)
        if (nbytes[i] > maxz) maxz = nbytes[i];
    unsigned char* zbuf = (unsigned char*)malloc((size_t)maxz + 16);
    float* blk = (float*)malloc(cbytes);
    if (!zbuf || !blk) {
        free(zbuf);
        free(blk);
        return -2;
    }
    int rc = 0;
    for (int i = 0; i < n_chunks && rc == 0; ++i) {
        const int f0 = origin[3 * i], r0 = origin[3 * i + 1], c0 = origin[3 * i + 2];
        const int nf = ext[0] - f0 < cdim[0] ? ext[0] - f0 : cdim[0];
        const int nr = ext[1] - r0 < cdim[1] ? ext[1] - r0 : cdim[1];
        const int nc = ext[2] - c0 < cdim[2] ? ext[2] - c0 : cdim[2];
        const int have = off[i] >= 0 && nbytes[i] > 0;
        if (have) {
            size_t got = 0;
            while (got < (size_t)nbytes[i]) {
                const ssize_t k = pread(fd, zbuf + got, (size_t)nbytes[i] - got, (off_t)(off[i] + (int64_t)got));
                if (k <= 0) {
                    rc = -3;
                    break;
                }
                got += (size_t)k;
            }
            if (rc) break;
            uLongf outlen = (uLongf)cbytes;
            if (uncompress((Bytef*)blk, &outlen, zbuf, (uLong)nbytes[i]) != Z_OK || outlen != cbytes) {
                rc = -4;
                break;
            }
        }
        for (int k = 0; k < nf; ++k) {
            const int df = fmap[f0 + k];
            if (df < 0) continue;
            for (int r = 0; r < nr; ++r) {
                float* d = dst + (int64_t)df * dst_frame_stride + (int64_t)(r0 + r) * dst_row_stride + c0;
                if (have) {
                    memcpy(d, blk + ((size_t)k * cdim[1] + r) * cdim[2], (size_t)nc * sizeof(float));
                } else {
                    for (int c = 0; c < nc; ++c) d[c] = fill; /* a chunk that was never written: the fill value */
                }
            }
        }
    }
    free(zbuf);
    free(blk);
    return rc;
}

int smaat_io_abi_version(void) { return 1; }
