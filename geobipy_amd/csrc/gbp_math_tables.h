// gbp_math_tables.h -- GENERATED (numpy longdouble / 60-digit decimal -> double); see DESIGN.md section 3.3.
// GBP_EXP2_64_LIST[j] = 2^(j/64);  GBP_SINCOS_64_LIST[2j] = sin(2 pi j / 64), [2j+1] = cos(2 pi j / 64).
#pragma once
#define GBP_EXP2_64_LIST \
    { \
    0x1.0000000000000p+0, \
        0x1.02c9a3e778061p+0, \
        0x1.059b0d3158574p+0, \
        0x1.0874518759bc8p+0, \
        0x1.0b5586cf9890fp+0, \
        0x1.0e3ec32d3d1a2p+0, \
        0x1.11301d0125b51p+0, \
        0x1.1429aaea92de0p+0, \
        0x1.172b83c7d517bp+0, \
        0x1.1a35beb6fcb75p+0, \
        0x1.1d4873168b9aap+0, \
        0x1.2063b88628cd6p+0, \
        0x1.2387a6e756238p+0, \
        0x1.26b4565e27cddp+0, \
        0x1.29e9df51fdee1p+0, \
        0x1.2d285a6e4030bp+0, \
        0x1.306fe0a31b715p+0, \
        0x1.33c08b26416ffp+0, \
        0x1.371a7373aa9cbp+0, \
        0x1.3a7db34e59ff7p+0, \
        0x1.3dea64c123422p+0, \
        0x1.4160a21f72e2ap+0, \
        0x1.44e086061892dp+0, \
        0x1.486a2b5c13cd0p+0, \
        0x1.4bfdad5362a27p+0, \
        0x1.4f9b2769d2ca7p+0, \
        0x1.5342b569d4f82p+0, \
        0x1.56f4736b527dap+0, \
        0x1.5ab07dd485429p+0, \
        0x1.5e76f15ad2148p+0, \
        0x1.6247eb03a5585p+0, \
        0x1.6623882552225p+0, \
        0x1.6a09e667f3bcdp+0, \
        0x1.6dfb23c651a2fp+0, \
        0x1.71f75e8ec5f74p+0, \
        0x1.75feb564267c9p+0, \
        0x1.7a11473eb0187p+0, \
        0x1.7e2f336cf4e62p+0, \
        0x1.82589994cce13p+0, \
        0x1.868d99b4492edp+0, \
        0x1.8ace5422aa0dbp+0, \
        0x1.8f1ae99157736p+0, \
        0x1.93737b0cdc5e5p+0, \
        0x1.97d829fde4e50p+0, \
        0x1.9c49182a3f090p+0, \
        0x1.a0c667b5de565p+0, \
        0x1.a5503b23e255dp+0, \
        0x1.a9e6b5579fdbfp+0, \
        0x1.ae89f995ad3adp+0, \
        0x1.b33a2b84f15fbp+0, \
        0x1.b7f76f2fb5e47p+0, \
        0x1.bcc1e904bc1d2p+0, \
        0x1.c199bdd85529cp+0, \
        0x1.c67f12e57d14bp+0, \
        0x1.cb720dcef9069p+0, \
        0x1.d072d4a07897cp+0, \
        0x1.d5818dcfba487p+0, \
        0x1.da9e603db3285p+0, \
        0x1.dfc97337b9b5fp+0, \
        0x1.e502ee78b3ff6p+0, \
        0x1.ea4afa2a490dap+0, \
        0x1.efa1bee615a27p+0, \
        0x1.f50765b6e4540p+0, \
        0x1.fa7c1819e90d8p+0, \
    }

#define GBP_SINCOS_64_LIST \
    { \
    0x0.0p+0, 0x1.0000000000000p+0, \
        0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1, \
        0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1, \
        0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, \
        0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, \
        0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1, \
        0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, \
        0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, \
        0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1, \
        0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, \
        0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, \
        0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2, \
        0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, \
        0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, \
        0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3, \
        0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4, \
        0x1.0000000000000p+0, 0x0.0p+0, \
        0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4, \
        0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3, \
        0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2, \
        0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2, \
        0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2, \
        0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1, \
        0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1, \
        0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1, \
        0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1, \
        0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1, \
        0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1, \
        0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1, \
        0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1, \
        0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1, \
        0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1, \
        0x0.0p+0, -0x1.0000000000000p+0, \
        -0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1, \
        -0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1, \
        -0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1, \
        -0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1, \
        -0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1, \
        -0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1, \
        -0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1, \
        -0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1, \
        -0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1, \
        -0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1, \
        -0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2, \
        -0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2, \
        -0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2, \
        -0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3, \
        -0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4, \
        -0x1.0000000000000p+0, 0x0.0p+0, \
        -0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4, \
        -0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3, \
        -0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, \
        -0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, \
        -0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2, \
        -0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, \
        -0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, \
        -0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1, \
        -0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, \
        -0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, \
        -0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1, \
        -0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, \
        -0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, \
        -0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1, \
        -0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1, \
    }

namespace gbp {
// reduction constants: ln2/64 and pi/32 split hi + lo; 64/ln2; 32/pi
constexpr double LN2_64_HI = 0x1.62e42fefa39efp-7, LN2_64_LO = 0x1.abc9e3b39803fp-62, INV_LN2_64 = 0x1.71547652b82fep+6;
constexpr double PI_32_HI = 0x1.921fb54442d18p-4, PI_32_LO = 0x1.1a62633145c07p-58, INV_PI_32 = 0x1.45f306dc9c883p+3;
}  // namespace gbp
