// se_formats.cpp -- the data formats on either side of the path: what a
// host needs to hand GPU-produced ciphertexts to the SEAL-side adapter, and to write keys the
// device side reads.  Host-only code, no kernels.
//
//   * SEAL ciphertext layout   /root/reference/adapter/fileops.cpp:492-538 (ct[i + j*n] = c0 prime j,
//                              ct[i + j*n + np*n] = c1 prime j, uint64 per coefficient)
//   * text lines               /root/reference/device/lib/util_print.h:229-245, 491-508
//                              ("name : { v0, v1, ... }"), order of device/test/api_tests.c:30-90
//                              (values line, then c0 line, c1 line per prime); parsed by
//                              adapter/fileops.h:220-287
//   * key files                sk_<n>.dat, pk{0,1}_ntt_<n>_<q>.dat (fileops.c:140-204,
//                              adapter/fileops.cpp:58-75,209-258)
#include <errno.h>
#include <stdio.h>
#include <string.h>

#include <string>

#include "../../include/seal_embedded_amd.h"
#include "se_context.h"

extern "C" {

void se_amd_pack_seal_ciphertext_host(const uint32_t *c0, const uint32_t *c1, size_t n, size_t np,
                                      uint64_t *out)
{
    for (size_t j = 0; j < np; j++)
        for (size_t i = 0; i < n; i++)
        {
            out[i + j * n]          = c0[j * n + i];
            out[i + j * n + np * n] = c1[j * n + i];
        }
}

// print_poly_full: "%s : { " then values joined by ", ", the last one followed by " " and "}\n"
size_t se_amd_format_poly_text(const char *name, const uint32_t *poly, size_t n, char *buf, size_t cap)
{
    size_t need = 0;
    auto put = [&](const char *s, size_t len) {
        if (buf && need + len < cap) memcpy(buf + need, s, len);
        need += len;
    };
    char tmp[32];
    put(name, strlen(name));
    put(" : { ", 5);
    for (size_t i = 0; i < n; i++)
    {
        int len = snprintf(tmp, sizeof(tmp), "%u", poly[i]);
        put(tmp, (size_t)len);
        if (i + 1 < n)
            put(", ", 2);
        else
            put(" ", 1);
    }
    put("}\n", 2);
    if (buf && cap) buf[need < cap ? need : cap - 1] = 0;
    return need;
}

// print_poly_flpt_full: same frame, values printed with "%0.2f"
size_t se_amd_format_values_text(const char *name, const float *v, size_t len, char *buf, size_t cap)
{
    size_t need = 0;
    auto put = [&](const char *s, size_t l) {
        if (buf && need + l < cap) memcpy(buf + need, s, l);
        need += l;
    };
    char tmp[64];
    put(name, strlen(name));
    put(" : { ", 5);
    for (size_t i = 0; i < len; i++)
    {
        int l = snprintf(tmp, sizeof(tmp), "%0.2f", (double)v[i]);
        put(tmp, (size_t)l);
        if (i + 1 < len)
            put(", ", 2);
        else
            put(" ", 1);
    }
    put("}\n", 2);
    if (buf && cap) buf[need < cap ? need : cap - 1] = 0;
    return need;
}

// One ciphertext in the text form the adapter's verify path reads: optional "v (cleartext)" line,
// then per prime a "c0" line and a "c1" line (api_tests.c:30-42,75-90).
int se_amd_write_ciphertext_text(const char *path, int append, const float *values, size_t vlen,
                                 const uint32_t *c0, const uint32_t *c1, size_t n, size_t np)
{
    if (!path || !c0 || !c1) return SE_ERR_INVALD_ARGUMENT;
    FILE *f = fopen(path, append ? "a" : "w");
    if (!f)
    {
        seamd::set_last_error(std::string("cannot open ") + path + ": " + strerror(errno));
        return SE_ERR_INVALD_ARGUMENT;
    }
    std::string line;
    if (values)
    {
        line.resize(se_amd_format_values_text("v (cleartext)", values, vlen, nullptr, 0) + 1);
        size_t l = se_amd_format_values_text("v (cleartext)", values, vlen, &line[0], line.size());
        fwrite(line.data(), 1, l, f);
    }
    for (size_t j = 0; j < np; j++)
    {
        const uint32_t *polys[2] = {c0 + j * n, c1 + j * n};
        const char *names[2]     = {"c0", "c1"};
        for (int k = 0; k < 2; k++)
        {
            line.resize(se_amd_format_poly_text(names[k], polys[k], n, nullptr, 0) + 1);
            size_t l = se_amd_format_poly_text(names[k], polys[k], n, &line[0], line.size());
            fwrite(line.data(), 1, l, f);
        }
    }
    fclose(f);
    return SE_SUCCESS;
}

static int write_exact(const std::string &path, const void *src, size_t bytes)
{
    FILE *f = fopen(path.c_str(), "wb");
    if (!f)
    {
        seamd::set_last_error("cannot create " + path + ": " + strerror(errno));
        return SE_ERR_INVALD_ARGUMENT;
    }
    size_t put = fwrite(src, 1, bytes, f);
    fclose(f);
    return put == bytes ? SE_SUCCESS : SE_ERR_UNKNOWN;
}

int se_amd_save_secret_key_file(const char *dir, size_t n, const uint8_t *sk_packed)
{
    if (!dir || !sk_packed) return SE_ERR_INVALD_ARGUMENT;
    char name[64];
    snprintf(name, sizeof(name), "/sk_%zu.dat", n);
    return write_exact(std::string(dir) + name, sk_packed, n / 4);
}

int se_amd_save_public_key_files(const char *dir, size_t n, size_t np, const uint32_t *q,
                                 const uint32_t *pk0, const uint32_t *pk1)
{
    if (!dir || !q || !pk0 || !pk1) return SE_ERR_INVALD_ARGUMENT;
    char name[96];
    for (size_t j = 0; j < np; j++)
    {
        snprintf(name, sizeof(name), "/pk0_ntt_%zu_%u.dat", n, q[j]);
        int rc = write_exact(std::string(dir) + name, pk0 + j * n, n * 4);
        if (rc) return rc;
        snprintf(name, sizeof(name), "/pk1_ntt_%zu_%u.dat", n, q[j]);
        rc = write_exact(std::string(dir) + name, pk1 + j * n, n * 4);
        if (rc) return rc;
    }
    return SE_SUCCESS;
}

}  // extern "C"
