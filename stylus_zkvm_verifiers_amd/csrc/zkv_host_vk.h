// Host-side context logic: the two hard-coded verification keys and the SHA-256 digest chain that
// `initialize` runs once per verifier instance.  No field arithmetic happens on the host.
//
// Reference: /root/reference/contracts/src/risc0/crypto.rs:16-89 (RISC Zero VK), :112-195 (tagged digests, VK digest),
// risc0/verifier.rs:58-76,128-144 (initialize, selector), risc0/config.rs (tags, SYSTEM_STATE_ZERO_DIGEST),
// sp1/crypto.rs:7-91 (SP1 VK, negated beta/gamma/delta), sp1/config.rs (VERIFIER_HASH, VERSION).
#pragma once
#include <stdint.h>
#include <string.h>
#include "zkv_verify.h"

namespace zkv {
namespace host {

struct VkHex { const char* alpha[2]; const char* beta[4]; const char* gamma[4]; const char* delta[4]; int n_ic; const char* ic[MAX_IC][2]; };

// words in the reference's order: G2 = x[0] (imaginary), x[1] (real), y[0] (imaginary), y[1] (real)
static const VkHex RISC0_VK = {
    {"2D4D9AA7E302D9DF41749D5507949D05DBEA33FBB16C643B22F599A2BE6DF2E2", "14BEDD503C37CEB061D8EC60209FE345CE89830A19230301F076CAFF004D1926"},
    {"0967032FCBF776D1AFC985F88877F182D38480A653F2DECAA9794CBC3BF3060C", "0E187847AD4C798374D0D6732BF501847DD68BC0E071241E0213BC7FC13DB7AB",
     "304CFBD1E08A704A99F5E847D93F8C3CAAFDDEC46B7A0D379DA69A4D112346A7", "1739C1B1A457A8C7313123D24D2F9192F896B7C63EEA05A9D57F06547AD0CEC8"},
    {"198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2", "1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED",
     "090689D0585FF075EC9E99AD690C3395BC4B313370B38EF355ACDADCD122975B", "12C85EA5DB8C6DEB4AAB71808DCB408FE3D1E7690C43D37B4CE6CC0166FA7DAA"},
    {"03B03CD5EFFA95AC9BEE94F1F5EF907157BDA4812CCF0B4C91F42BB629F83A1C", "1AA085FF28179A12D922DBA0547057CCAAE94B9D69CFAA4E60401FEA7F3E0333",
     "110C10134F200B19F6490846D518C9AEA868366EFB7228CA5C91D2940D030762", "1E60F31FCBF757E837E867178318832D0B2D74D59E2FEA1C7142DF187D3FC6D3"},
    6,
    {{"12AC9A25DCD5E1A832A9061A082C15DD1D61AA9C4D553505739D0F5D65DC3BE4", "025AA744581EBE7AD91731911C898569106FF5A2D30F3EEE2B23C60EE980ACD4"},
     {"0707B920BC978C02F292FAE2036E057BE54294114CCC3C8769D883F688A1423F", "2E32A094B7589554F7BC357BF63481ACD2D55555C203383782A4650787FF6642"},
     {"0BCA36E2CBE6394B3E249751853F961511011C7148E336F4FD974644850FC347", "2EDE7C9ACF48CF3A3729FA3D68714E2A8435D4FA6DB8F7F409C153B1FCDF9B8B"},
     {"1B8AF999DBFBB3927C091CC2AAF201E488CBACC3E2C6B6FB5A25F9112E04F2A7", "2B91A26AA92E1B6F5722949F192A81C850D586D81A60157F3E9CF04F679CCCD6"},
     {"2B5F494ED674235B8AC1750BDFD5A7615F002D4A1DCEFEDDD06EDA5A076CCD0D", "2FE520AD2020AAB9CBBA817FCBB9A863B8A76FF88F14F912C5E71665B2AD5E82"},
     {"0F1C3C0D5D9DA0FA03666843CDE4E82E869BA5252FCE3C25D5940320B1C4D493", "214BFCFF74F425F6FE8C0D07B307482D8BC8BB2F3608F68287AA01BD0B69E809"}}};

static const VkHex SP1_VK = {
    {"2D4D9AA7E302D9DF41749D5507949D05DBEA33FBB16C643B22F599A2BE6DF2E2", "14BEDD503C37CEB061D8EC60209FE345CE89830A19230301F076CAFF004D1926"},
    {"0967032FCBF776D1AFC985F88877F182D38480A653F2DECAA9794CBC3BF3060C", "0E187847AD4C798374D0D6732BF501847DD68BC0E071241E0213BC7FC13DB7AB",
     "001752A100A72FDF1E5A5D6EA841CC20EC838BCCFCF7BD559E79F1C9C759B6A0", "192A8CC13CD9F762871F21E43451C6CA9EEAB2CB2987C4E366A185C25DAC2E7F"},
    {"198E9393920D483A7260BFB731FB5D25F1AA493335A9E71297E485B7AEF312C2", "1800DEEF121F1E76426A00665E5C4479674322D4F75EDADD46DEBD5CD992F6ED",
     "275DC4A288D1AFB3CBB1AC09187524C7DB36395DF7BE3B99E673B13A075A65EC", "1D9BEFCD05A5323E6DA4D435F3B617CDB3AF83285C2DF711EF39C01571827F9D"},
    {"1CC7CB8DE715675F21F01ECC9B46D236E0865E0CC020024521998269845F74E6", "03FF41F4BA0C37FE2CAF27354D28E4B8F83D3B76777A63B327D736BFFB0122ED",
     "01909CD7827E0278E6B60843A4ABC7B111D7F8B2725CD5902A6B20DA7A2938FB", "192BD3274441670227B4F69A44005B8711266E474227C6439CA25CA8E1EC1FC2"},
    3,
    {{"26091E1CAFB0AD8A4EA0A694CD3743EBF524779233DB734C451D28B58AA9758E", "009FF50A6B8B11C3CA6FDB2690A124F8CE25489FEFA65A3E782E7BA70B66690E"},
     {"061C3FD0FD3DA25D2607C227D090CCA750ED36C6EC878755E537C1C48951FB4C", "0FA17AE9C2033379DF7B5C65EFF0E107055E9A273E6119A212DD09EB51707219"},
     {"04EAB241388A79817FE0E0E2EAD0B2EC4FFDEC51A16028DEE020634FD129E71C", "07236256D21C60D02F0BDBF95CFF83E03EA9E16FCA56B18D5544B0889A65C1F5"},
     {nullptr, nullptr}, {nullptr, nullptr}, {nullptr, nullptr}}};

static const uint8_t SYSTEM_STATE_ZERO_DIGEST[32] = {       // risc0/config.rs:5-10
    0xa3, 0xac, 0xc2, 0x71, 0x17, 0x41, 0x89, 0x96, 0x34, 0x0b, 0x84, 0xe5, 0xa9, 0x0f, 0x3e, 0xf4,
    0xc4, 0x9d, 0x22, 0xc7, 0x9e, 0x44, 0xaa, 0xd8, 0x22, 0xec, 0x9c, 0x31, 0x3e, 0x1e, 0xb8, 0xe2};
static const uint8_t SP1_VERIFIER_HASH[32] = {              // sp1/config.rs:4-9
    0xa4, 0x59, 0x4c, 0x59, 0xbb, 0xc1, 0x42, 0xf3, 0xb8, 0x1c, 0x3e, 0xcb, 0x7f, 0x50, 0xa7, 0xc3,
    0x4b, 0xc9, 0xaf, 0x7c, 0x4c, 0x44, 0x4b, 0x5d, 0x48, 0xb7, 0x95, 0x42, 0x7e, 0x28, 0x59, 0x13};
static const char SP1_VERSION[] = "v5.0.0";                  // sp1/config.rs:3

inline void hex32(uint8_t out[32], const char* h) {
    for (int i = 0; i < 32; i++) {
        int v = 0;
        for (int k = 0; k < 2; k++) {
            char c = h[2 * i + k];
            v = v * 16 + (c >= 'a' ? c - 'a' + 10 : c >= 'A' ? c - 'A' + 10 : c - '0');
        }
        out[i] = (uint8_t)v;
    }
}
inline void be_to_limbs(uint32_t limbs[8], const uint8_t be[32]) {
    for (int i = 0; i < 8; i++)
        limbs[7 - i] = ((uint32_t)be[4 * i] << 24) | ((uint32_t)be[4 * i + 1] << 16) | ((uint32_t)be[4 * i + 2] << 8) | be[4 * i + 3];
}
inline void hex_to_limbs(uint32_t limbs[8], const char* h) { uint8_t b[32]; hex32(b, h); be_to_limbs(limbs, b); }

inline void sha256_host(const uint8_t* msg, size_t len, uint8_t out[32]) {
    uint32_t h[8];
    sha256_bytes(msg, len, h);
    for (int i = 0; i < 8; i++) { out[4 * i] = h[i] >> 24; out[4 * i + 1] = h[i] >> 16; out[4 * i + 2] = h[i] >> 8; out[4 * i + 3] = h[i]; }
}

inline void fill_vk_common(VkRaw& r, const VkHex& v) {
    memset(&r, 0, sizeof r);
    hex_to_limbs(r.alpha[0], v.alpha[0]); hex_to_limbs(r.alpha[1], v.alpha[1]);
    // reference order is (x_im, x_re, y_im, y_re); VkRaw wants (x_re, x_im, y_re, y_im)
    const int perm[4] = {1, 0, 3, 2};
    for (int k = 0; k < 4; k++) {
        hex_to_limbs(r.beta[k], v.beta[perm[k]]); hex_to_limbs(r.gamma[k], v.gamma[perm[k]]); hex_to_limbs(r.delta[k], v.delta[perm[k]]);
    }
    r.n_ic = (uint32_t)v.n_ic;
    for (int i = 0; i < v.n_ic; i++) { hex_to_limbs(r.ic[i][0], v.ic[i][0]); hex_to_limbs(r.ic[i][1], v.ic[i][1]); }
}
// risc0: signals [control_root_0, control_root_1, claim_lo, claim_hi, bn254_control_id] (verifier.rs:173-179);
// 0, 1, 4 are per-context, 2 and 3 (128-bit halves) vary per proof.
inline void fill_vk_risc0(VkRaw& r, const uint8_t cr0[16], const uint8_t cr1[16], const uint8_t control_id[32]) {
    fill_vk_common(r, RISC0_VK);
    uint8_t w[32];
    memset(w, 0, 32); memcpy(w + 16, cr0, 16); be_to_limbs(r.fixed_scalar[1], w);
    memset(w, 0, 32); memcpy(w + 16, cr1, 16); be_to_limbs(r.fixed_scalar[2], w);
    be_to_limbs(r.fixed_scalar[5], control_id);
    r.is_fixed[1] = r.is_fixed[2] = r.is_fixed[5] = 1;
    r.n_var = 2; r.var_ic[0] = 3; r.var_ic[1] = 4; r.var_windows[0] = 16; r.var_windows[1] = 16;
}
// sp1: signals [program_vkey, hash(public_values)] both per proof (sp1/verifier.rs:85-86)
inline void fill_vk_sp1(VkRaw& r) {
    fill_vk_common(r, SP1_VK);
    r.n_var = 2; r.var_ic[0] = 1; r.var_ic[1] = 2; r.var_windows[0] = 32; r.var_windows[1] = 32;
}

// Arbitrary verification key (Groth16Verifier::verify_proof_with_key is generic over `vk`, common/groth16.rs:23-31):
// words = alpha1.x alpha1.y | beta2.x[0] x[1] y[0] y[1] | gamma2 (4) | delta2 (4) | ic[i].x ic[i].y ..., 32-byte big-endian,
// G2 words in the reference's (imaginary, real) order.  Every signal is per-proof.
inline void fill_vk_generic(VkRaw& r, const uint8_t* words, uint32_t n_ic) {
    memset(&r, 0, sizeof r);
    be_to_limbs(r.alpha[0], words); be_to_limbs(r.alpha[1], words + 32);
    const int perm[4] = {1, 0, 3, 2};
    for (int k = 0; k < 4; k++) {
        be_to_limbs(r.beta[k], words + 64 + 32 * perm[k]);
        be_to_limbs(r.gamma[k], words + 192 + 32 * perm[k]);
        be_to_limbs(r.delta[k], words + 320 + 32 * perm[k]);
    }
    r.n_ic = n_ic;
    for (uint32_t i = 0; i < n_ic; i++) { be_to_limbs(r.ic[i][0], words + 448 + 64 * i); be_to_limbs(r.ic[i][1], words + 480 + 64 * i); }
    r.n_var = n_ic - 1;
    for (uint32_t b = 0; b + 1 < n_ic; b++) { r.var_ic[b] = b + 1; r.var_windows[b] = 32; }
}

// ---- risc0 digest chain (crypto.rs:95-195, verifier.rs:128-144)
inline void split_digest(const uint8_t d[32], uint8_t lo[16], uint8_t hi[16]) {
    uint8_t rev[32];
    for (int i = 0; i < 32; i++) rev[i] = d[31 - i];
    memcpy(lo, rev + 16, 16); memcpy(hi, rev, 16);
}
inline void tagged_struct(const uint8_t tag[32], const uint8_t* down, int n, uint8_t out[32]) {
    uint8_t buf[32 + 32 * 8 + 2];
    memcpy(buf, tag, 32); memcpy(buf + 32, down, 32 * (size_t)n);
    uint16_t v = (uint16_t)(n << 8);
    buf[32 + 32 * n] = (uint8_t)(v >> 8); buf[33 + 32 * n] = (uint8_t)(v & 0xff);
    sha256_host(buf, 34 + 32 * (size_t)n, out);
}
inline void risc0_vk_digest(uint8_t out[32]) {
    const VkHex& vk = RISC0_VK;
    uint8_t icd[6][32], buf[128], tag[32], down[64], cur[32], parts[5][32];
    for (int i = 0; i < 6; i++) { hex32(buf, vk.ic[i][0]); hex32(buf + 32, vk.ic[i][1]); sha256_host(buf, 64, icd[i]); }
    hex32(buf, vk.alpha[0]); hex32(buf + 32, vk.alpha[1]); sha256_host(buf, 64, parts[0]);
    for (int k = 0; k < 4; k++) hex32(buf + 32 * k, vk.beta[k]);
    sha256_host(buf, 128, parts[1]);
    for (int k = 0; k < 4; k++) hex32(buf + 32 * k, vk.gamma[k]);
    sha256_host(buf, 128, parts[2]);
    for (int k = 0; k < 4; k++) hex32(buf + 32 * k, vk.delta[k]);
    sha256_host(buf, 128, parts[3]);
    sha256_host((const uint8_t*)"risc0_groth16.VerifyingKey.IC", 29, tag);
    memset(cur, 0, 32);
    for (int i = 5; i >= 0; i--) { memcpy(down, icd[i], 32); memcpy(down + 32, cur, 32); tagged_struct(tag, down, 2, cur); }
    memcpy(parts[4], cur, 32);
    sha256_host((const uint8_t*)"risc0_groth16.VerifyingKey", 26, tag);
    tagged_struct(tag, &parts[0][0], 5, out);
}
inline void risc0_selector(const uint8_t control_root[32], const uint8_t control_id[32], uint8_t sel[4]) {
    uint8_t buf[130], h[32];
    sha256_host((const uint8_t*)"risc0.Groth16ReceiptVerifierParameters", 38, buf);
    memcpy(buf + 32, control_root, 32);
    for (int i = 0; i < 32; i++) buf[64 + i] = control_id[31 - i];
    risc0_vk_digest(buf + 96);
    buf[128] = 3; buf[129] = 0;
    sha256_host(buf, 130, h);
    memcpy(sel, h, 4);
}
inline void risc0_consts(Risc0Consts& k) {
    uint8_t tag[32], blk[64];
    sha256_host((const uint8_t*)"risc0.Output", 12, tag);
    for (int i = 0; i < 8; i++) k.tag_output[i] = load_be32(tag + 4 * i);
    sha256_host((const uint8_t*)"risc0.ReceiptClaim", 18, blk);
    memset(blk + 32, 0, 32);
    uint32_t w[16];
    for (int i = 0; i < 16; i++) w[i] = load_be32(blk + 4 * i);
    sha256_init(k.claim_mid);
    sha256_compress(k.claim_mid, w);
    for (int i = 0; i < 8; i++) k.post[i] = load_be32(SYSTEM_STATE_ZERO_DIGEST + 4 * i);
}

}  // namespace host
}  // namespace zkv
