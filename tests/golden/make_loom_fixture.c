/* make_loom_fixture.c -- writes tests/golden/loompy_v2.loom and tests/golden/loompy_v3.loom WITHOUT this repository's own
 * writer (velocyto_amd.loom_io.write_loom): a C program straight on libhdf5, laying the files out the way loompy does
 * for the output of `velocyto run` (reference: velocyto/commands/_run.py:283-297 builds the layers and attributes,
 * loompy.create / LoomConnection write them; velocyto/constants.py:11 fixes the layer dtype; velocyto/analysis.py:56-67
 * and 2314-2342 read them back).  Purpose: pin velocyto_amd.loom_io.read_loom to files it did not write itself.
 *
 * Layout reproduced (loompy 2.0.x = LOOM_SPEC_VERSION 2.0.1, and loompy 3 = spec 3.0.0):
 *   /matrix            float32 (genes, cells), chunked (64, 64), gzip level 2, maxshape unlimited
 *   /layers/<name>     spliced, unspliced, ambiguous: uint16 (v2 file) / uint32 (v3 file), same chunking + gzip
 *   /row_attrs/<name>  Gene, Accession, Chromosome, Strand (strings), Start, End (int64)
 *   /col_attrs/<name>  CellID (strings), Clusters (int64), _X, _Y (float64; v2) / TSNE (cells, 2) float64 (v3)
 *   /row_graphs, /col_graphs   empty groups
 *   strings            v2: fixed-length ASCII ("S<n>", null-padded), as loompy 2 normalises them
 *                      v3: variable-length UTF-8
 *   file attributes    v2: HDF5 attributes on "/" (LOOM_SPEC_VERSION = "2.0.1", CreationDate)
 *                      v3: datasets under /attrs (LOOM_SPEC_VERSION = "3.0.0" scalar variable-length string, CreationDate)
 *
 * The numbers follow closed formulas (below) that tests/test_loom_io.py re-evaluates in NumPy; no reference data involved.
 *
 * build + run (this container; HDF5 1.10.6 from /opt/conda):
 *   gcc tests/golden/make_loom_fixture.c -I/opt/conda/include -L/opt/conda/lib -lhdf5 -Wl,-rpath,/opt/conda/lib -o /tmp/make_loom_fixture
 *   /tmp/make_loom_fixture tests/golden
 */
#include <hdf5.h>
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#define NG 23
#define NC 17

static unsigned spliced(int g, int c) { return (unsigned)((g * 7 + c * 3) % 11) * ((g + c) % 3 != 0) + (g == 5 && c == 4 ? 40000u : 0u) + (g == 20 && c == 16 ? 300u : 0u); }
static unsigned unspliced(int g, int c) { return (unsigned)((g * 5 + c * 2) % 7) * ((g * c) % 4 != 1); }
static unsigned ambiguous(int g, int c) { return (unsigned)((g + 2 * c) % 5 == 0); }

static void die(const char *what) { fprintf(stderr, "make_loom_fixture: %s failed\n", what); exit(1); }
#define CHK(x, what) do { if ((x) < 0) die(what); } while (0)

static hid_t chunked_gzip(int rank)
{
    hid_t p = H5Pcreate(H5P_DATASET_CREATE);
    hsize_t ch[2] = {64, 64};
    CHK(H5Pset_chunk(p, rank, ch), "H5Pset_chunk");
    CHK(H5Pset_deflate(p, 2), "H5Pset_deflate");
    return p;
}

static void put_matrix(hid_t parent, const char *name, hid_t file_type, hid_t mem_type, const void *buf)
{
    hsize_t dims[2] = {NG, NC}, maxd[2] = {H5S_UNLIMITED, H5S_UNLIMITED};
    hid_t sp = H5Screate_simple(2, dims, maxd), pl = chunked_gzip(2);
    hid_t d = H5Dcreate2(parent, name, file_type, sp, H5P_DEFAULT, pl, H5P_DEFAULT);
    CHK(d, name);
    CHK(H5Dwrite(d, mem_type, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), name);
    H5Dclose(d); H5Pclose(pl); H5Sclose(sp);
}

static void put_numeric(hid_t parent, const char *name, hid_t type, int rank, const hsize_t *dims, const void *buf)
{
    hid_t sp = H5Screate_simple(rank, dims, NULL);
    hid_t d = H5Dcreate2(parent, name, type, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHK(d, name);
    CHK(H5Dwrite(d, type, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), name);
    H5Dclose(d); H5Sclose(sp);
}

/* n strings: fixed-length ASCII (width = longest, null padded) or variable-length UTF-8 */
static void put_strings(hid_t parent, const char *name, int n, const char **vals, int vlen)
{
    hsize_t dims[1] = {(hsize_t)n};
    hid_t sp = H5Screate_simple(1, dims, NULL), tp = H5Tcopy(H5T_C_S1), d;
    if (vlen) {
        CHK(H5Tset_size(tp, H5T_VARIABLE), "H5Tset_size");
        CHK(H5Tset_cset(tp, H5T_CSET_UTF8), "H5Tset_cset");
        d = H5Dcreate2(parent, name, tp, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        CHK(d, name);
        CHK(H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, vals), name);
    } else {
        size_t w = 1;
        for (int i = 0; i < n; ++i) if (strlen(vals[i]) > w) w = strlen(vals[i]);
        char *buf = calloc((size_t)n, w);
        for (int i = 0; i < n; ++i) memcpy(buf + (size_t)i * w, vals[i], strlen(vals[i]));
        CHK(H5Tset_size(tp, w), "H5Tset_size");
        CHK(H5Tset_strpad(tp, H5T_STR_NULLPAD), "H5Tset_strpad");
        d = H5Dcreate2(parent, name, tp, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        CHK(d, name);
        CHK(H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, buf), name);
        free(buf);
    }
    H5Dclose(d); H5Tclose(tp); H5Sclose(sp);
}

static void put_scalar_vlen(hid_t parent, const char *name, const char *val)
{
    hid_t sp = H5Screate(H5S_SCALAR), tp = H5Tcopy(H5T_C_S1);
    H5Tset_size(tp, H5T_VARIABLE); H5Tset_cset(tp, H5T_CSET_UTF8);
    hid_t d = H5Dcreate2(parent, name, tp, sp, H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    CHK(d, name);
    CHK(H5Dwrite(d, tp, H5S_ALL, H5S_ALL, H5P_DEFAULT, &val), name);
    H5Dclose(d); H5Tclose(tp); H5Sclose(sp);
}

static void put_root_attr(hid_t f, const char *name, const char *val)
{
    hid_t sp = H5Screate(H5S_SCALAR), tp = H5Tcopy(H5T_C_S1);
    H5Tset_size(tp, strlen(val) + 1);
    hid_t a = H5Acreate2(f, name, tp, sp, H5P_DEFAULT, H5P_DEFAULT);
    CHK(a, name);
    CHK(H5Awrite(a, tp, val), name);
    H5Aclose(a); H5Tclose(tp); H5Sclose(sp);
}

static void write_file(const char *path, int v3)
{
    hid_t f = H5Fcreate(path, H5F_ACC_TRUNC, H5P_DEFAULT, H5P_DEFAULT);
    CHK(f, path);
    static float m[NG][NC];
    static uint16_t l16[3][NG][NC];
    static uint32_t l32[3][NG][NC];
    for (int g = 0; g < NG; ++g)
        for (int c = 0; c < NC; ++c) {
            unsigned s = spliced(g, c), u = unspliced(g, c), a = ambiguous(g, c);
            m[g][c] = (float)(s + u + a);
            l16[0][g][c] = (uint16_t)s; l16[1][g][c] = (uint16_t)u; l16[2][g][c] = (uint16_t)a;
            l32[0][g][c] = s; l32[1][g][c] = u; l32[2][g][c] = a;
        }
    put_matrix(f, "matrix", H5T_IEEE_F32LE, H5T_NATIVE_FLOAT, m);
    hid_t lay = H5Gcreate2(f, "layers", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    const char *names[3] = {"spliced", "unspliced", "ambiguous"};
    for (int i = 0; i < 3; ++i) {
        if (v3) put_matrix(lay, names[i], H5T_STD_U32LE, H5T_NATIVE_UINT32, l32[i]);
        else put_matrix(lay, names[i], H5T_STD_U16LE, H5T_NATIVE_UINT16, l16[i]);
    }
    H5Gclose(lay);
    /* ---- row attributes */
    hid_t ra = H5Gcreate2(f, "row_attrs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char gene[NG][32], acc[NG][32], chrom[NG][8];
    const char *pg[NG], *pa[NG], *pc[NG], *ps[NG];
    int64_t start[NG], end[NG];
    for (int g = 0; g < NG; ++g) {
        if (v3 && g == 2) snprintf(gene[g], sizeof gene[g], "G\xc3\xa8ne_%d", g);        /* UTF-8 e-grave: only a vlen UTF-8 file can hold it */
        else snprintf(gene[g], sizeof gene[g], "Gene_%d", g * g);
        snprintf(acc[g], sizeof acc[g], "ENSMUSG%011d", 1000 + 37 * g);
        snprintf(chrom[g], sizeof chrom[g], "%d", 1 + g % 19);
        pg[g] = gene[g]; pa[g] = acc[g]; pc[g] = chrom[g]; ps[g] = (g % 2) ? "-" : "+";
        start[g] = 100000LL * g + 17; end[g] = start[g] + 1500 + 13 * g;
    }
    put_strings(ra, "Gene", NG, pg, v3);
    put_strings(ra, "Accession", NG, pa, v3);
    put_strings(ra, "Chromosome", NG, pc, v3);
    put_strings(ra, "Strand", NG, ps, v3);
    hsize_t dg[1] = {NG};
    put_numeric(ra, "Start", H5T_NATIVE_INT64, 1, dg, start);
    put_numeric(ra, "End", H5T_NATIVE_INT64, 1, dg, end);
    H5Gclose(ra);
    /* ---- column attributes */
    hid_t ca = H5Gcreate2(f, "col_attrs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
    char cid[NC][40];
    const char *pi[NC];
    int64_t clusters[NC];
    double x[NC], y[NC], ts[NC][2];
    for (int c = 0; c < NC; ++c) {
        snprintf(cid[c], sizeof cid[c], "sample1:%cACGT%04dx", 'A' + c % 4, c * 31);
        pi[c] = cid[c];
        clusters[c] = c % 3;
        x[c] = 0.5 * c - 3.25; y[c] = 1.0 / (1 + c);
        ts[c][0] = x[c]; ts[c][1] = y[c];
    }
    put_strings(ca, "CellID", NC, pi, v3);
    hsize_t dc[2] = {NC, 2};
    put_numeric(ca, "Clusters", H5T_NATIVE_INT64, 1, dc, clusters);
    if (v3) put_numeric(ca, "TSNE", H5T_NATIVE_DOUBLE, 2, dc, ts);
    else { put_numeric(ca, "_X", H5T_NATIVE_DOUBLE, 1, dc, x); put_numeric(ca, "_Y", H5T_NATIVE_DOUBLE, 1, dc, y); }
    H5Gclose(ca);
    H5Gclose(H5Gcreate2(f, "row_graphs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    H5Gclose(H5Gcreate2(f, "col_graphs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT));
    if (v3) {
        hid_t at = H5Gcreate2(f, "attrs", H5P_DEFAULT, H5P_DEFAULT, H5P_DEFAULT);
        put_scalar_vlen(at, "LOOM_SPEC_VERSION", "3.0.0");
        put_scalar_vlen(at, "CreationDate", "20180808T120000.000000Z");
        H5Gclose(at);
    } else {
        put_root_attr(f, "LOOM_SPEC_VERSION", "2.0.1");
        put_root_attr(f, "CreationDate", "2018/08/08 12:00:00");
    }
    CHK(H5Fclose(f), "H5Fclose");
}

int main(int argc, char **argv)
{
    const char *dir = argc > 1 ? argv[1] : ".";
    char path[1024];
    snprintf(path, sizeof path, "%s/loompy_v2.loom", dir);
    write_file(path, 0);
    snprintf(path, sizeof path, "%s/loompy_v3.loom", dir);
    write_file(path, 1);
    printf("wrote %s/loompy_v2.loom and %s/loompy_v3.loom (%d genes x %d cells)\n", dir, dir, NG, NC);
    return 0;
}
