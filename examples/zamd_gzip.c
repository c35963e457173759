/*
 * zamd_gzip — compress or decompress files with the gz* layer of libzamd_z.so,
 * the way the reference's test/minigzip.c drives its own library: the input is
 * read in large pieces and handed to gzwrite(), or read back through gzread().
 *
 *   zamd_gzip [-d] [-1..-9] [-f|-h|-r] file...     file -> file.gz (or file.gz -> file with -d); the input is kept
 *
 * Build: cc -Iinclude examples/zamd_gzip.c -Lzlib_amd -lzamd_z -lzamd -Wl,-rpath,$PWD/zlib_amd
 */
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

#include "zamd_zlib.h"

#define PIECE (64u << 20)

static int fail(const char *what, const char *name)
{
    fprintf(stderr, "zamd_gzip: %s: %s\n", name, what);
    return 1;
}

static int pack(const char *name, const char *mode, unsigned char *buf)
{
    char out[4096];
    if (snprintf(out, sizeof out, "%s.gz", name) >= (int)sizeof out) return fail("name too long", name);
    FILE *in = fopen(name, "rb");
    if (!in) return fail("cannot open", name);
    gzFile g = gzopen(out, mode);
    if (!g) { fclose(in); return fail("cannot create", out); }
    size_t n;
    while ((n = fread(buf, 1, PIECE, in)) > 0)
        if (gzwrite(g, buf, (unsigned)n) != (int)n) { int e; fail(gzerror(g, &e), out); gzclose(g); fclose(in); return 1; }
    fclose(in);
    if (gzclose(g) != Z_OK) return fail("close failed", out);
    return 0;
}

static int unpack(const char *name, unsigned char *buf)
{
    size_t l = strlen(name);
    char out[4096];
    if (l < 4 || strcmp(name + l - 3, ".gz") != 0 || l >= sizeof out) return fail("no .gz suffix", name);
    memcpy(out, name, l - 3); out[l - 3] = 0;
    gzFile g = gzopen(name, "rb");
    if (!g) return fail("cannot open", name);
    FILE *o = fopen(out, "wb");
    if (!o) { gzclose(g); return fail("cannot create", out); }
    int n;
    while ((n = gzread(g, buf, PIECE)) > 0)
        if (fwrite(buf, 1, (size_t)n, o) != (size_t)n) { fclose(o); gzclose(g); return fail("write failed", out); }
    int e = 0;
    const char *msg = gzerror(g, &e);
    if (n < 0 || (e != Z_OK && e != Z_STREAM_END)) { fail(msg, name); fclose(o); gzclose(g); return 1; }
    fclose(o);
    return gzclose(g) == Z_OK ? 0 : fail("close failed", name);
}

int main(int argc, char **argv)
{
    int decompress = 0, rc = 0, want_index = 0;
    char mode[8] = "wb6";
    unsigned char *buf = (unsigned char *)malloc(PIECE);
    if (!buf) return 2;
    for (int i = 1; i < argc; i++) {
        const char *a = argv[i];
        if (a[0] == '-' && a[1] && !a[2]) {
            if (a[1] == 'd') decompress = 1;
            else if (a[1] >= '1' && a[1] <= '9') mode[2] = a[1];
            else if (a[1] == 'f') strcpy(mode + 3, "f");
            else if (a[1] == 'h') strcpy(mode + 3, "h");
            else if (a[1] == 'r') strcpy(mode + 3, "R");
            else if (a[1] == 'i') want_index = 1;         /* append the chunk index (this library's gz layer only) */
            else { fprintf(stderr, "usage: zamd_gzip [-d] [-1..-9] [-f|-h|-r] [-i] file...\n"); return 2; }
            continue;
        }
        if (want_index && !strchr(mode, 'I')) strcat(mode, "I");
        rc |= decompress ? unpack(a, buf) : pack(a, mode, buf);
    }
    free(buf);
    return rc;
}
