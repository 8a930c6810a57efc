/* Sanitizer driver for the CPU oracle (test infrastructure): built together with zkv_oracle.c under
 * -fsanitize=address,undefined by tests/test_oracle_sanitize.py.
 *   driver risc0 <control_root> <control_id> <seal> <image_id> <journal_digest>
 *   driver sp1 <vkey> <public_values> <proof>
 *   driver call_risc0 <control_root> <control_id> <calldata>      (control_root "-" = un-initialised verifier)
 *   driver call_sp1 <calldata>
 * prints the status code (verify modes) or "reverted returndata-hex" (call modes). */
#include <stdint.h>
#include <stdio.h>
#include <stdlib.h>
#include <string.h>

typedef struct zkvo_risc0 zkvo_risc0;
zkvo_risc0* zkvo_risc0_new(void);
void zkvo_risc0_free(zkvo_risc0*);
int zkvo_risc0_initialize(zkvo_risc0*, const uint8_t*, const uint8_t*);
int zkvo_risc0_verify(const zkvo_risc0*, const uint8_t*, size_t, const uint8_t*, const uint8_t*, uint8_t*);
int zkvo_sp1_verify_proof(const uint8_t*, const uint8_t*, size_t, const uint8_t*, size_t, uint8_t*);
int zkvo_risc0_eth_call(const zkvo_risc0*, const uint8_t*, size_t, uint8_t*, size_t*, int*);
int zkvo_sp1_eth_call(const uint8_t*, size_t, uint8_t*, size_t*, int*);

static uint8_t* unhex(const char* h, size_t* n) {
    size_t len = strlen(h) / 2;
    uint8_t* b = (uint8_t*)malloc(len ? len : 1);
    for (size_t i = 0; i < len; i++) { unsigned v; sscanf(h + 2 * i, "%2x", &v); b[i] = (uint8_t)v; }
    *n = len;
    return b;
}

int main(int argc, char** argv) {
    uint8_t recv[4] = {0, 0, 0, 0};
    size_t n1, n2, n3, n4, n5;
    if (argc == 7 && !strcmp(argv[1], "risc0")) {
        uint8_t *cr = unhex(argv[2], &n1), *cid = unhex(argv[3], &n2), *seal = unhex(argv[4], &n3), *im = unhex(argv[5], &n4), *jd = unhex(argv[6], &n5);
        zkvo_risc0* v = zkvo_risc0_new();
        zkvo_risc0_initialize(v, cr, cid);
        printf("%d\n", zkvo_risc0_verify(v, seal, n3, im, jd, recv));
        zkvo_risc0_free(v); free(cr); free(cid); free(seal); free(im); free(jd);
        return 0;
    }
    if (argc == 5 && !strcmp(argv[1], "sp1")) {
        uint8_t *vk = unhex(argv[2], &n1), *pv = unhex(argv[3], &n2), *pr = unhex(argv[4], &n3);
        printf("%d\n", zkvo_sp1_verify_proof(vk, pv, n2, pr, n3, recv));
        free(vk); free(pv); free(pr);
        return 0;
    }
    if ((argc == 5 && !strcmp(argv[1], "call_risc0")) || (argc == 3 && !strcmp(argv[1], "call_sp1"))) {
        uint8_t ret[96]; size_t rl = 0; int st = 0, rev;
        if (argc == 5) {
            zkvo_risc0* v = zkvo_risc0_new();
            if (strcmp(argv[2], "-")) { uint8_t *cr = unhex(argv[2], &n1), *cid = unhex(argv[3], &n2); zkvo_risc0_initialize(v, cr, cid); free(cr); free(cid); }
            uint8_t* cd = unhex(argv[4], &n3);
            uint8_t* exact = (uint8_t*)malloc(n3 ? n3 : 1);       /* exactly-sized copy: any over-read trips ASan */
            memcpy(exact, cd, n3);
            rev = zkvo_risc0_eth_call(v, exact, n3, ret, &rl, &st);
            free(exact); free(cd); zkvo_risc0_free(v);
        } else {
            uint8_t* cd = unhex(argv[2], &n3);
            uint8_t* exact = (uint8_t*)malloc(n3 ? n3 : 1);
            memcpy(exact, cd, n3);
            rev = zkvo_sp1_eth_call(exact, n3, ret, &rl, &st);
            free(exact); free(cd);
        }
        printf("%d ", rev);
        for (size_t i = 0; i < rl; i++) printf("%02x", ret[i]);
        printf("\n");
        return 0;
    }
    return 2;
}
