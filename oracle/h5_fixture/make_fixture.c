/* Writes the HDF5 fixture of tests/golden/ with the REAL HDF5 library (libhdf5 1.10.6, found under /opt/conda in the
 * build container), issuing the library calls h5py makes for /root/reference/create_datasets.py:31-61:
 *   f.create_group("train" | "test")
 *   group.create_dataset("images", shape=(1, T, H, W), maxshape=(None, T, H, W), dtype="float32",
 *                        compression="gzip", compression_opts=9)          -> chunked, h5py's guessed chunk shape,
 *                                                                            deflate level 9, no shuffle
 *   group.create_dataset("timestamps", shape=(1, T, 1), maxshape=(None, T, 1), dtype=vlen str, gzip 9)
 *   dataset.resize(total, axis=0); dataset[idx] = imgs                     (:84-92, one sample per write)
 * h5py opens files with libver "earliest": superblock 0, version-1 object headers, symbol-table groups, version-1
 * B-tree chunk index -- the library defaults used here.
 *
 *   make_fixture OUT.h5 TRAIN.bin n_train TEST.bin n_test T H W  c0 c1 c2 c3
 * TRAIN.bin / TEST.bin: raw little-endian float32 [n][T][H][W]; c0..c3: chunk shape (oracle/h5_fixture/gen_h5_fixture.py
 * computes it with a restatement of h5py's guess_chunk).  TEST INFRASTRUCTURE ONLY (never shipped, never on the GPU box).
 */
#include <hdf5.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define CHECK(x)                                                         \
    do {                                                                 \
        if ((x) < 0) {                                                   \
            fprintf(stderr, "HDF5 call failed: %s (line %d)\n", #x, __LINE__); \
            exit(2);                                                     \
        }                                                                \
    } while (0)

static float* slurp(const char* path, size_t count) {
    FILE* f = fopen(path, "rb");
    if (!f) { perror(path); exit(2); }
    float* p = (float*)malloc(count * sizeof(float));
    if (fread(p, sizeof(float), count, f) != count) { fprintf(stderr, "short read: %s\n", path); exit(2); }
    fclose(f);
    return p;
}

static void write_split(hid_t file, const char* name, const float* data, hsize_t n, hsize_t T, hsize_t H, hsize_t W,
                        const hsize_t chunk[4]) {
    hid_t grp = H5Gcreate2(file, name, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHECK(grp);
    /* images */
    hsize_t dims[4] = {1, T, H, W}, maxd[4] = {H5S_UNLIMITED, T, H, W};
    hid_t sp = H5Screate_simple(4, dims, maxd);
    hid_t dcpl = H5Pcreate(H5P_DATASET_CREATE);
    CHECK(H5Pset_chunk(dcpl, 4, chunk));
    CHECK(H5Pset_deflate(dcpl, 9));
    hid_t ds = H5Dcreate2(grp, "images", H5T_IEEE_F32LE, sp, H5P_DEFAULT, dcpl, H5P_DEFAULT);
    CHECK(ds);
    hsize_t ext[4] = {n, T, H, W};
    CHECK(H5Dset_extent(ds, ext)); /* image_dataset.resize(total_sequences, axis=0) */
    hid_t fsp = H5Dget_space(ds);
    hsize_t cnt[4] = {1, T, H, W};
    hid_t msp = H5Screate_simple(4, cnt, NULL);
    for (hsize_t i = 0; i < n; ++i) { /* image_dataset[idx] = imgs */
        hsize_t start[4] = {i, 0, 0, 0};
        CHECK(H5Sselect_hyperslab(fsp, H5S_SELECT_SET, start, NULL, cnt, NULL));
        CHECK(H5Dwrite(ds, H5T_NATIVE_FLOAT, msp, fsp, H5P_DEFAULT, data + i * T * H * W));
    }
    H5Sclose(msp); H5Sclose(fsp); H5Dclose(ds); H5Pclose(dcpl); H5Sclose(sp);
    /* timestamps: variable-length strings, gzip 9 (a reader of /images must walk past this object) */
    hsize_t tdims[3] = {1, T, 1}, tmax[3] = {H5S_UNLIMITED, T, 1}, tchunk[3] = {1, T, 1};
    hid_t tsp = H5Screate_simple(3, tdims, tmax);
    hid_t tcpl = H5Pcreate(H5P_DATASET_CREATE);
    CHECK(H5Pset_chunk(tcpl, 3, tchunk));
    CHECK(H5Pset_deflate(tcpl, 9));
    hid_t st = H5Tcopy(H5T_C_S1);
    CHECK(H5Tset_size(st, H5T_VARIABLE));
    CHECK(H5Tset_cset(st, H5T_CSET_UTF8));
    hid_t tds = H5Dcreate2(grp, "timestamps", st, tsp, H5P_DEFAULT, tcpl, H5P_DEFAULT);
    CHECK(tds);
    hsize_t text[3] = {n, T, 1};
    CHECK(H5Dset_extent(tds, text));
    hid_t tfsp = H5Dget_space(tds);
    hsize_t tcnt[3] = {1, T, 1};
    hid_t tmsp = H5Screate_simple(3, tcnt, NULL);
    char** strs = (char**)malloc(T * sizeof(char*));
    for (hsize_t i = 0; i < n; ++i) {
        for (hsize_t t = 0; t < T; ++t) {
            strs[t] = (char*)malloc(40);
            snprintf(strs[t], 40, "2016-01-%02d %02d:%02d:00;%s", (int)(i % 28) + 1, (int)(t * 5 / 60), (int)(t * 5 % 60), name);
        }
        hsize_t start[3] = {i, 0, 0};
        CHECK(H5Sselect_hyperslab(tfsp, H5S_SELECT_SET, start, NULL, tcnt, NULL));
        CHECK(H5Dwrite(tds, st, tmsp, tfsp, H5P_DEFAULT, strs));
        for (hsize_t t = 0; t < T; ++t) free(strs[t]);
    }
    free(strs);
    H5Sclose(tmsp); H5Sclose(tfsp); H5Dclose(tds); H5Tclose(st); H5Pclose(tcpl); H5Sclose(tsp);
    H5Gclose(grp);
}

int main(int argc, char** argv) {
    if (argc != 13) {
        fprintf(stderr, "usage: %s OUT.h5 TRAIN.bin n_train TEST.bin n_test T H W c0 c1 c2 c3\n", argv[0]);
        return 1;
    }
    const hsize_t ntr = strtoull(argv[3], 0, 10), nte = strtoull(argv[5], 0, 10);
    const hsize_t T = strtoull(argv[6], 0, 10), H = strtoull(argv[7], 0, 10), W = strtoull(argv[8], 0, 10);
    hsize_t chunk[4];
    for (int i = 0; i < 4; ++i) chunk[i] = strtoull(argv[9 + i], 0, 10);
    float* tr = slurp(argv[2], ntr * T * H * W);
    float* te = slurp(argv[4], nte * T * H * W);
    hid_t file = H5Fcreate(argv[1], H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT); /* h5py.File(filename, "w") */
    CHECK(file);
    write_split(file, "train", tr, ntr, T, H, W, chunk);
    write_split(file, "test", te, nte, T, H, W, chunk);
    CHECK(H5Fclose(file));
    unsigned maj, min, rel;
    H5get_libversion(&maj, &min, &rel);
    printf("wrote %s with HDF5 %u.%u.%u\n", argv[1], maj, min, rel);
    return 0;
}
