// lt_kernels.hip -- gfx950 (CDNA4) kernels of the photon-transport hot path.
//
// Design:
//   * one wavefront lane per live photon; photon state lives in VGPRs;
//   * persistent threads: the grid is sized to the resident capacity of the
//     chip and every wave loops, pulling PACKETS of photon ids from one global
//     counter (1 returning atomic per 64 photons) -- divergent path lengths are
//     absorbed by refilling dead lanes, not by launching more threads;
//   * dead lanes are found with a wave ballot and refilled by ballot rank
//     (v_mbcnt), four at a time (a refill costs the whole wave ~140 instructions);
//   * rocRAND XORWOW state per lane, re-seeded per photon from (seed, id) and
//     warmed up by 8 discarded outputs, so a photon's uniforms do not depend
//     on which lane / wave / GPU traces it;
//   * the kernel body lives in lt_walk_kernel.inc and is instantiated twice:
//     walk_kernel (slabs) and walk_kernel_q (meshes: BVH queries batched per
//     wave, the one divergent block expensive enough to be worth waiting for);
//   * media, layer and BVH/triangle tables are staged once per workgroup into
//     LDS (all lanes read the same few entries: LDS broadcast);
//   * deposits: consecutive same-voxel deposits of a lane are merged in
//     registers; the surviving records go either to the grid with no-return
//     global atomics (f32 / f64 / u64 fixed point) or -- default for slabs --
//     into a coalesced deposit log that lt_logtally.hip partitions by grid tile
//     and reduces in LDS (the memory-side atomic unit, ~16e9 requests/s, was
//     the limiter; HBM streaming bandwidth is not); rare per-photon events go
//     to LDS counters;
//   * f64 arithmetic (the reference's dtype) with lean -ln / sincos / sqrt /
//     quotient primitives for the hot loop's restricted argument ranges.
// No MFMA: there is no dense contraction anywhere on this path.
//
// Reference citations: S/ = LightTransportSimulator/light_transport/src/.
#include <hip/hip_runtime.h>

#define ROCRAND_DETAIL_BM_NOT_IN_STATE
#include <rocrand/rocrand_xorwow.h>
#include <rocrand/rocrand_uniform.h>

#include "lt_internal.hpp"

// the tail split hands a photon's generator over as six words (SurvD::rng = {d, x[5]}): true only while the Box-Muller
// cache is compiled out of the state (ROCRAND_DETAIL_BM_NOT_IN_STATE above) and the header keeps this layout
static_assert(sizeof(rocrand_state_xorwow) == 6 * sizeof(uint32_t), "rocrand_state_xorwow is not {d, x[5]}: SurvD::rng and the tail-split hand-over assume it");

namespace ltk {

#define LT_DEV __device__ __forceinline__
// a loop invariant of the walk in walk precision R, read from the kernel arguments (WalkParams P / MarchGrid G): the f64
// value or its f32 mirror -- a compile-time choice
#define LT_PV(name, k) (sizeof(R) == 8 ? (R)P.name[k] : (R)P.f32.name[k])
#define LT_GV(name, k) (sizeof(R) == 8 ? (R)G.name[k] : (R)G.name##32[k])

// ---------------------------------------------------------------------------
// precision traits
// ---------------------------------------------------------------------------
// Lean f64 primitives for the walk's restricted argument ranges.  OCML's general-purpose f64 log / sincospi /
// sqrt / division cost 98 / 71 / 22 / 12 VALU instructions each because they cover denormals, huge arguments and
// correct rounding; the walk only ever needs -ln(xi) for xi in [2^-53, 1], sin/cos of a turn fraction in (0, 1],
// sqrt on [0, 1] and quotients of well-scaled numbers.  These versions are accurate to ~1-2 ulp (checked against
// the host libm in tests/test_gpu_parity.py through lt_eval) at 35 / 38 / 9 / 8 instructions.
// a * b + k with the constant k in a scalar register pair: ONE v_fma_f64.  Left to itself the compiler parks
// polynomial coefficients in VGPRs and spends a v_mov_b64 + v_fmac_f64 per Horner step (the VOP2 form accumulates
// into its addend, and an f64 literal operand only carries the high dword; gfx9's VOP3 takes no literal at all).
// The pair is written by two s_mov_b32 INSIDE the asm, into s[100:101], which the asm clobbers: handed in as an "s"
// operand the ~40 constants of the walk's polynomials are loop invariants, all live through the hot loop at once
// (80 SGPRs on top of the kernel's pointers and scalars), and the register allocator answered by spilling scalars into
// VGPR lanes and restoring them with v_readlane -- VALU instructions -- every iteration (tens per photon-step once the
// loop's other invariants had moved from VGPRs to kernel arguments).  Two scalar moves per use cost the VALU nothing.
#ifdef LT_K_IN_SGPR     /* A/B builds: the round-3 form, constants as "s" operands (loop invariants in scalar registers) */
LT_DEV double fma_k(double a, double b, double k) { double r; asm("v_fma_f64 %0, %1, %2, %3" : "=v"(r) : "v"(a), "v"(b), "s"(k)); return r; }
LT_DEV double fma_mk(double a, double k, double c) { return __builtin_fma(a, k, c); }
LT_DEV double mul_k(double a, double k) { return a * k; }
LT_DEV double fma_kk(double x, double k) { double r; asm("v_fma_f64 %0, %1, %2, %2" : "=v"(r) : "v"(x), "s"(k)); return r; }
#else
template <unsigned long long BITS> LT_DEV double fma_kb(double a, double b)
{
    double r;
    asm("s_mov_b32 s100, %3\n\ts_mov_b32 s101, %4\n\tv_fma_f64 %0, %1, %2, s[100:101]"
        : "=v"(r) : "v"(a), "v"(b), "i"((int)(unsigned)(BITS & 0xffffffffull)), "i"((int)(unsigned)(BITS >> 32)) : "s100", "s101");
    return r;
}
#define fma_k(a, b, k) fma_kb<__builtin_bit_cast(unsigned long long, (double)(k))>((a), (b))
// a * k + c, the constant as the multiplier
template <unsigned long long BITS> LT_DEV double fma_mkb(double a, double c)
{
    double r;
    asm("s_mov_b32 s100, %3\n\ts_mov_b32 s101, %4\n\tv_fma_f64 %0, %1, s[100:101], %2"
        : "=v"(r) : "v"(a), "v"(c), "i"((int)(unsigned)(BITS & 0xffffffffull)), "i"((int)(unsigned)(BITS >> 32)) : "s100", "s101");
    return r;
}
#define fma_mk(a, k, c) fma_mkb<__builtin_bit_cast(unsigned long long, (double)(k))>((a), (c))
// a * k
template <unsigned long long BITS> LT_DEV double mul_kb(double a)
{
    double r;
    asm("s_mov_b32 s100, %2\n\ts_mov_b32 s101, %3\n\tv_mul_f64 %0, %1, s[100:101]"
        : "=v"(r) : "v"(a), "i"((int)(unsigned)(BITS & 0xffffffffull)), "i"((int)(unsigned)(BITS >> 32)) : "s100", "s101");
    return r;
}
#define mul_k(a, k) mul_kb<__builtin_bit_cast(unsigned long long, (double)(k))>((a))
// x * k + k, one v_fma_f64 with k read twice from the same scalar pair: the (0, 1] uniform conversions of
// rocrand_uniform.h:97-109 (k = 2^-32 / 2^-53), which otherwise cost two v_mov_b32 to seed a v_fmac_f64
template <unsigned long long BITS> LT_DEV double fma_kkb(double x)
{
    double r;
    asm("s_mov_b32 s100, %2\n\ts_mov_b32 s101, %3\n\tv_fma_f64 %0, %1, s[100:101], s[100:101]"
        : "=v"(r) : "v"(x), "i"((int)(unsigned)(BITS & 0xffffffffull)), "i"((int)(unsigned)(BITS >> 32)) : "s100", "s101");
    return r;
}
#define fma_kk(x, k) fma_kkb<__builtin_bit_cast(unsigned long long, (double)(k))>((x))
#endif
LT_DEV double fast_div(double a, double b)
{
    // ONE Newton step on the seed (-> ~52 bits) is enough here: the residual correction below squares the error once more
    double r = __builtin_amdgcn_rcp(b);                 // v_rcp_f64: ~26 good bits
    r = __builtin_fma(__builtin_fma(-b, r, 1.0), r, r);
    const double q = a * r;
    return __builtin_fma(__builtin_fma(-b, q, a), r, q); // one residual correction
}
LT_DEV double sqrt_core(double x, double y)              // y ~ 1/sqrt(x) to ~26 bits (v_rsq_f64)
{
    // one coupled Newton round (26 -> ~52 bits), then the residual step squares the error again: correctly rounded but for rare ties
    double g = x * y, h = 0.5 * y;
    const double r = __builtin_fma(-h, g, 0.5);
    g = __builtin_fma(g, r, g); h = __builtin_fma(h, r, h);
    const double d = __builtin_fma(-g, g, x);
    return __builtin_fma(d, h, g);
}
LT_DEV double sqrt01(double x)                           // x in [0, 1]; 0 (and a rounding-negative x) gives 0
{
    const double xc = __builtin_fmax(x, 0.0);
    // the seed of 0 is taken at 1e-300 (finite); every product with x = 0 is then an exact 0, no select needed
    return sqrt_core(xc, __builtin_amdgcn_rsq(__builtin_fmax(xc, 1e-300)));
}
LT_DEV double sqrt_pos(double x) { return sqrt_core(x, __builtin_amdgcn_rsq(x)); }   // x > 0 known
LT_DEV double rsqrt_pos(double x)                        // 1 / sqrt(x), x > 0 well scaled: v_rsq_f64 + 2 Newton steps
{
    double y = __builtin_amdgcn_rsq(x);
    double e = __builtin_fma(-(x * y), y, 1.0);
    y = __builtin_fma(0.5 * y, e, y);
    e = __builtin_fma(-(x * y), y, 1.0);
    return __builtin_fma(0.5 * y, e, y);
}
// -ln(x) and sin / cos(2 pi x) of the step (free path, azimuth) by TABLE + SHORT POLYNOMIAL: a step spends 17 instructions
// less on the azimuth and a dozen less on the logarithm than with the table-free forms they replace (two degree-7 / 8
// polynomials + a quadrant rotation by selects; a quotient + a ten-term series).  kWalkMathTab: 97 x {1 / j, ln(j / 128)},
// j = 96..192, then 64 x {sin, cos}(2 pi i / 64) -- correctly rounded values, 2576 bytes; the walk kernels stage them in LDS
// (one 16-byte read per use), the per-function kernel reads them from global memory.  Accuracy as before: 2.5e-16 relative
// for the logarithm (the cell around m = 1 has ln c = 0 and r = m - 1 exactly, so -ln x -> 1 - x keeps its relative accuracy),
// 1.2e-16 absolute for sin / cos; exact at the quarter turns.
__device__ const double kWalkMathTab[(97 + 64) * 2] = {
    // {1 / j, ln(j / 128)}, j = 96 .. 192
    0x1.5555555555555p-7, -0x1.269621134db92p-2, 0x1.51d07eae2f815p-7, -0x1.1bf99635a6b95p-2, 0x1.4e5e0a72f0539p-7, -0x1.1178e8227e47cp-2,
    0x1.4afd6a052bf5bp-7, -0x1.07138604d5862p-2, 0x1.47ae147ae147bp-7, -0x1.f991c6cb3b379p-3, 0x1.446f86562d9fbp-7, -0x1.e530effe71012p-3,
    0x1.4141414141414p-7, -0x1.d1037f2655e7bp-3, 0x1.3e22cbce4a902p-7, -0x1.bd087383bd8adp-3, 0x1.3b13b13b13b14p-7, -0x1.a93ed3c8ad9e3p-3,
    0x1.3813813813814p-7, -0x1.95a5adcf7017fp-3, 0x1.3521cfb2b78c1p-7, -0x1.823c16551a3c2p-3, 0x1.323e34a2b10bfp-7, -0x1.6f0128b756abcp-3,
    0x1.2f684bda12f68p-7, -0x1.5bf406b543db2p-3, 0x1.2c9fb4d812ca0p-7, -0x1.4913d8333b561p-3, 0x1.29e4129e4129ep-7, -0x1.365fcb0159016p-3,
    0x1.27350b8812735p-7, -0x1.23d712a49c202p-3, 0x1.2492492492492p-7, -0x1.1178e8227e47cp-3, 0x1.21fb78121fb78p-7, -0x1.fe89139dbd566p-4,
    0x1.1f7047dc11f70p-7, -0x1.da727638446a2p-4, 0x1.1cf06ada2811dp-7, -0x1.b6ac88dad5b1cp-4, 0x1.1a7b9611a7b96p-7, -0x1.9335e5d594989p-4,
    0x1.1811811811812p-7, -0x1.700d30aeac0e1p-4, 0x1.15b1e5f75270dp-7, -0x1.4d3115d207eacp-4, 0x1.135c81135c811p-7, -0x1.2aa04a44717a5p-4,
    0x1.1111111111111p-7, -0x1.08598b59e3a07p-4, 0x1.0ecf56be69c90p-7, -0x1.ccb73cdddb2ccp-5, 0x1.0c9714fbcda3bp-7, -0x1.894aa149fb343p-5,
    0x1.0a6810a6810a7p-7, -0x1.466aed42de3eap-5, 0x1.0842108421084p-7, -0x1.0415d89e74444p-5, 0x1.0624dd2f1a9fcp-7, -0x1.8492528c8cabfp-6,
    0x1.0410410410410p-7, -0x1.0205658935847p-6, 0x1.0204081020408p-7, -0x1.010157588de71p-7, 0x1.0000000000000p-7, 0x0.0p+0,
    0x1.fc07f01fc07f0p-8, 0x1.fe02a6b106789p-8, 0x1.f81f81f81f820p-8, 0x1.fc0a8b0fc03e4p-7, 0x1.f44659e4a4271p-8, 0x1.7b91b07d5b11bp-6,
    0x1.f07c1f07c1f08p-8, 0x1.f829b0e783300p-6, 0x1.ecc07b301ecc0p-8, 0x1.39e87b9febd60p-5, 0x1.e9131abf0b767p-8, 0x1.77458f632dcfcp-5,
    0x1.e573ac901e574p-8, 0x1.b42dd711971bfp-5, 0x1.e1e1e1e1e1e1ep-8, 0x1.f0a30c01162a6p-5, 0x1.de5d6e3f8868ap-8, 0x1.16536eea37ae1p-4,
    0x1.dae6076b981dbp-8, 0x1.341d7961bd1d1p-4, 0x1.d77b654b82c34p-8, 0x1.51b073f06183fp-4, 0x1.d41d41d41d41dp-8, 0x1.6f0d28ae56b4cp-4,
    0x1.d0cb58f6ec074p-8, 0x1.8c345d6319b21p-4, 0x1.cd85689039b0bp-8, 0x1.a926d3a4ad563p-4, 0x1.ca4b3055ee191p-8, 0x1.c5e548f5bc743p-4,
    0x1.c71c71c71c71cp-8, 0x1.e27076e2af2e6p-4, 0x1.c3f8f01c3f8f0p-8, 0x1.fec9131dbeabbp-4, 0x1.c0e070381c0e0p-8, 0x1.0d77e7cd08e59p-3,
    0x1.bdd2b899406f7p-8, 0x1.1b72ad52f67a0p-3, 0x1.bacf914c1bad0p-8, 0x1.29552f81ff523p-3, 0x1.b7d6c3dda338bp-8, 0x1.371fc201e8f74p-3,
    0x1.b4e81b4e81b4fp-8, 0x1.44d2b6ccb7d1ep-3, 0x1.b2036406c80d9p-8, 0x1.526e5e3a1b438p-3, 0x1.af286bca1af28p-8, 0x1.5ff3070a793d4p-3,
    0x1.ac5701ac5701bp-8, 0x1.6d60fe719d21dp-3, 0x1.a98ef606a63bep-8, 0x1.7ab890210d909p-3, 0x1.a6d01a6d01a6dp-8, 0x1.87fa06520c911p-3,
    0x1.a41a41a41a41ap-8, 0x1.9525a9cf456b4p-3, 0x1.a16d3f97a4b02p-8, 0x1.a23bc1fe2b563p-3, 0x1.9ec8e951033d9p-8, 0x1.af3c94e80bff3p-3,
    0x1.9c2d14ee4a102p-8, 0x1.bc286742d8cd6p-3, 0x1.999999999999ap-8, 0x1.c8ff7c79a9a22p-3, 0x1.970e4f80cb872p-8, 0x1.d5c216b4fbb91p-3,
    0x1.948b0fcd6e9e0p-8, 0x1.e27076e2af2e6p-3, 0x1.920fb49d0e229p-8, 0x1.ef0adcbdc5936p-3, 0x1.8f9c18f9c18fap-8, 0x1.fb9186d5e3e2bp-3,
    0x1.8d3018d3018d3p-8, 0x1.0402594b4d041p-2, 0x1.8acb90f6bf3aap-8, 0x1.0a324e27390e3p-2, 0x1.886e5f0abb04ap-8, 0x1.1058bf9ae4ad5p-2,
    0x1.8618618618618p-8, 0x1.1675cababa60ep-2, 0x1.83c977ab2beddp-8, 0x1.1c898c16999fbp-2, 0x1.8181818181818p-8, 0x1.22941fbcf7966p-2,
    0x1.7f405fd017f40p-8, 0x1.2895a13de86a3p-2, 0x1.7d05f417d05f4p-8, 0x1.2e8e2bae11d31p-2, 0x1.7ad2208e0ecc3p-8, 0x1.347dd9a987d55p-2,
    0x1.78a4c8178a4c8p-8, 0x1.3a64c556945eap-2, 0x1.767dce434a9b1p-8, 0x1.404308686a7e4p-2, 0x1.745d1745d1746p-8, 0x1.4618bc21c5ec2p-2,
    0x1.724287f46debcp-8, 0x1.4be5f957778a1p-2, 0x1.702e05c0b8170p-8, 0x1.51aad872df82dp-2, 0x1.6e1f76b4337c7p-8, 0x1.5767717455a6cp-2,
    0x1.6c16c16c16c17p-8, 0x1.5d1bdbf5809cap-2, 0x1.6a13cd1537290p-8, 0x1.62c82f2b9c795p-2, 0x1.6816816816817p-8, 0x1.686c81e9b14afp-2,
    0x1.661ec6a5122f9p-8, 0x1.6e08eaa2ba1e4p-2, 0x1.642c8590b2164p-8, 0x1.739d7f6bbd007p-2, 0x1.623fa77016240p-8, 0x1.792a55fdd47a2p-2,
    0x1.6058160581606p-8, 0x1.7eaf83b82afc3p-2, 0x1.5e75bb8d015e7p-8, 0x1.842d1da1e8b17p-2, 0x1.5c9882b931057p-8, 0x1.89a3386c1425bp-2,
    0x1.5ac056b015ac0p-8, 0x1.8f11e873662c7p-2, 0x1.58ed2308158edp-8, 0x1.947941c2116fbp-2, 0x1.571ed3c506b3ap-8, 0x1.99d958117e08bp-2,
    0x1.5555555555555p-8, 0x1.9f323ecbf984cp-2,
    // {sin, cos}(2 pi i / 64), i = 0 .. 63
    0x0.0p+0, 0x1.0000000000000p+0, 0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1, 0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1,
    0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, 0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, 0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1,
    0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, 0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, 0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1,
    0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, 0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, 0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2,
    0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, 0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, 0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3,
    0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4, 0x1.0000000000000p+0, 0x0.0p+0, 0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4,
    0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3, 0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2, 0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2,
    0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2, 0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1, 0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1,
    0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1, 0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1, 0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1,
    0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1, 0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1, 0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1,
    0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1, 0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1, 0x0.0p+0, -0x1.0000000000000p+0,
    -0x1.917a6bc29b42cp-4, -0x1.fd88da3d12526p-1, -0x1.8f8b83c69a60bp-3, -0x1.f6297cff75cb0p-1, -0x1.294062ed59f06p-2, -0x1.e9f4156c62ddap-1,
    -0x1.87de2a6aea963p-2, -0x1.d906bcf328d46p-1, -0x1.e2b5d3806f63bp-2, -0x1.c38b2f180bdb1p-1, -0x1.1c73b39ae68c8p-1, -0x1.a9b66290ea1a3p-1,
    -0x1.44cf325091dd6p-1, -0x1.8bc806b151741p-1, -0x1.6a09e667f3bcdp-1, -0x1.6a09e667f3bcdp-1, -0x1.8bc806b151741p-1, -0x1.44cf325091dd6p-1,
    -0x1.a9b66290ea1a3p-1, -0x1.1c73b39ae68c8p-1, -0x1.c38b2f180bdb1p-1, -0x1.e2b5d3806f63bp-2, -0x1.d906bcf328d46p-1, -0x1.87de2a6aea963p-2,
    -0x1.e9f4156c62ddap-1, -0x1.294062ed59f06p-2, -0x1.f6297cff75cb0p-1, -0x1.8f8b83c69a60bp-3, -0x1.fd88da3d12526p-1, -0x1.917a6bc29b42cp-4,
    -0x1.0000000000000p+0, 0x0.0p+0, -0x1.fd88da3d12526p-1, 0x1.917a6bc29b42cp-4, -0x1.f6297cff75cb0p-1, 0x1.8f8b83c69a60bp-3,
    -0x1.e9f4156c62ddap-1, 0x1.294062ed59f06p-2, -0x1.d906bcf328d46p-1, 0x1.87de2a6aea963p-2, -0x1.c38b2f180bdb1p-1, 0x1.e2b5d3806f63bp-2,
    -0x1.a9b66290ea1a3p-1, 0x1.1c73b39ae68c8p-1, -0x1.8bc806b151741p-1, 0x1.44cf325091dd6p-1, -0x1.6a09e667f3bcdp-1, 0x1.6a09e667f3bcdp-1,
    -0x1.44cf325091dd6p-1, 0x1.8bc806b151741p-1, -0x1.1c73b39ae68c8p-1, 0x1.a9b66290ea1a3p-1, -0x1.e2b5d3806f63bp-2, 0x1.c38b2f180bdb1p-1,
    -0x1.87de2a6aea963p-2, 0x1.d906bcf328d46p-1, -0x1.294062ed59f06p-2, 0x1.e9f4156c62ddap-1, -0x1.8f8b83c69a60bp-3, 0x1.f6297cff75cb0p-1,
    -0x1.917a6bc29b42cp-4, 0x1.fd88da3d12526p-1
};
constexpr int kLnTabEntries = 97, kScTabEntries = 64;
constexpr size_t kWalkMathTabBytes = sizeof(double) * 2 * (kLnTabEntries + kScTabEntries);
LT_DEV double neg_log_tab(double x, const double* T)     // -ln(x), x in [2^-53, 1]; T = kWalkMathTab (global or its LDS copy)
{
    int e = __builtin_amdgcn_frexp_exp(x);
    double m = __builtin_amdgcn_frexp_mant(x);           // [0.5, 1)
    if (m < 0.75) { m += m; e -= 1; }                    // -> [0.75, 1.5)
    const double t = m * 128.0;                          // exact
    const double qf = __builtin_rint(t);                 // the cell's centre c = qf / 128, qf = 96 .. 192
    const double d = t - qf;                             // exact: 128 (m - c)
    const double2 Tj = *reinterpret_cast<const double2*>(T + 2 * ((int)qf - 96));
    const double r = d * Tj.x;                           // (m - c) / c, |r| <= 1 / 192
    // ln(1 + r) = r - r^2/2 + ... + r^7/7; the truncated tail is < 2e-18 relative to r
    double p = 1.0 / 7.0;
    p = fma_k(p, r, -1.0 / 6.0); p = fma_k(p, r, 1.0 / 5.0); p = fma_k(p, r, -0.25); p = fma_k(p, r, 1.0 / 3.0);
    p = __builtin_fma(p, r, -0.5);
    const double lnm = Tj.y + __builtin_fma(r * r, p, r);
    const double de = (double)e;
    // ln 2 split so that e * hi is exact for |e| <= 53
    return -fma_mk(de, 6.93147180369123816490e-01, fma_mk(de, 1.90821492927058770002e-10, lnm));
}
LT_DEV void sincos_turn_tab(double xi, const double* T, double* sn, double* cs)  // sin, cos of 2*pi*xi, xi in (0, 1]; T = kWalkMathTab
{
    const double t = 64.0 * xi;                          // exact
    const double qf = __builtin_rint(t);
    const double a = mul_k(t - qf, 9.8174770424681038702e-02);   // (t - qf) exact; 2 pi / 64; |a| <= pi / 64
    const double z = a * a;
    double ps = -1.9841269841269841270e-04;              // -1/7!   (tail a^9 / 9! < 5e-18)
    ps = fma_k(ps, z, 8.3333333333333333333e-03);   //  1/5!
    ps = fma_k(ps, z, -1.6666666666666666667e-01);  // -1/3!
    const double sd = __builtin_fma(a * z, ps, a);
    double pc = 2.4801587301587301587e-05;               //  1/8!   (tail z^5 / 10! < 1e-20)
    pc = fma_k(pc, z, -1.3888888888888888889e-03);  // -1/6!
    pc = fma_k(pc, z, 4.1666666666666666667e-02);   //  1/4!
    pc = __builtin_fma(pc, z, -0.5);
    const double cd = __builtin_fma(pc, z, 1.0);
    const double2 Ti = *reinterpret_cast<const double2*>(T + 2 * (kLnTabEntries + ((int)qf & (kScTabEntries - 1))));
    *sn = __builtin_fma(Ti.x, cd, Ti.y * sd);            // angle addition: the cell's sin / cos turned by the remainder
    *cs = __builtin_fma(Ti.y, cd, -(Ti.x * sd));
}

// sin, cos of x for |x| <= pi (the renderers' concentric disk map: theta in [-pi/4, 3 pi/4]): reduction by multiples of pi/2
// (Cody-Waite, two terms: exact for |k| <= 2) and the polynomials above.  OCML's general routines carry a Payne-Hanek
// reduction for huge arguments and cost the render kernels tens of VGPRs; within 1 ulp of them here.
LT_DEV void sincos_small_f64(double x, double* sn, double* cs)
{
    const double kf = __builtin_rint(x * 0.63661977236758134308);          // x * 2 / pi
    double a = __builtin_fma(-kf, 1.57079632679489655800e+00, x);           // pi/2, high part
    a = __builtin_fma(-kf, 6.12323399573676603587e-17, a);                  //       low part
    const double z = a * a;
    double ps = -7.6471637318198164759e-13;
    ps = fma_k(ps, z, 1.6059043836821614599e-10);
    ps = fma_k(ps, z, -2.5052108385441718775e-08);
    ps = fma_k(ps, z, 2.7557319223985890653e-06);
    ps = fma_k(ps, z, -1.9841269841269841270e-04);
    ps = fma_k(ps, z, 8.3333333333333333333e-03);
    ps = fma_k(ps, z, -1.6666666666666666667e-01);
    const double s = __builtin_fma(a * z, ps, a);
    double pc = 4.7794773323873852974e-14;
    pc = fma_k(pc, z, -1.1470745597729724714e-11);
    pc = fma_k(pc, z, 2.0876756987868098979e-09);
    pc = fma_k(pc, z, -2.7557319223985890653e-07);
    pc = fma_k(pc, z, 2.4801587301587301587e-05);
    pc = fma_k(pc, z, -1.3888888888888888889e-03);
    pc = fma_k(pc, z, 4.1666666666666666667e-02);
    pc = __builtin_fma(pc, z, -0.5);
    const double c = __builtin_fma(pc, z, 1.0);
    const int q = (int)kf & 3;
    const double s1 = (q & 1) ? c : s, c1 = (q & 1) ? s : c;
    *sn = (q & 2) ? -s1 : s1;
    *cs = ((q + 1) & 2) ? -c1 : c1;
}

// The same two functions from the RAW 32-bit draw k (the uniform is (k + 1) 2^-32): exponent, table cell and remainder come out
// of integer shifts instead of frexp / ldexp / rint in f64 -- ~5 f64 operations less each, and no conversion of the uniform.
LT_DEV double neg_log_raw(unsigned k, const double* T)
{
    const unsigned n = k + 1u;                           // 0 stands for 2^32: the uniform is 1, -ln is 0
    const int lz = __clz((int)n);
    const unsigned nm = n << (lz & 31);                  // mantissa with bit 31 set: m = nm / 2^31 in [1, 2)
    const unsigned hi = nm >= 0xC0000000u ? 1u : 0u;     // m >= 1.5: halve it -> [0.75, 1.5)
    const unsigned sh = 24u + hi;                        // t = 128 m' = nm / 2^sh
    const unsigned j = ((nm >> (sh - 1u)) + 1u) >> 1;    // rint(t): the cell's centre, 96 .. 192
    const int dint = (int)(nm - (j << sh));              // (t - j) 2^sh, exact (modular arithmetic covers j 2^sh = 2^32)
    const double d = __builtin_ldexp((double)dint, -(int)sh);
    const double2 Tj = *reinterpret_cast<const double2*>(T + 2 * ((int)j - 96));
    const double r = d * Tj.x;
    double p = 1.0 / 7.0;
    p = fma_k(p, r, -1.0 / 6.0); p = fma_k(p, r, 1.0 / 5.0); p = fma_k(p, r, -0.25); p = fma_k(p, r, 1.0 / 3.0);
    p = __builtin_fma(p, r, -0.5);
    const double lnm = Tj.y + __builtin_fma(r * r, p, r);
    const double de = (double)((int)hi - 1 - lz);        // x = m' 2^e
    const double v = -fma_mk(de, 6.93147180369123816490e-01, fma_mk(de, 1.90821492927058770002e-10, lnm));
    return n == 0u ? 0.0 : v;
}
LT_DEV void sincos_turn_raw(unsigned k, const double* T, double* sn, double* cs)
{
    const unsigned n = k + 1u;                           // (n = 0 stands for a full turn: cell 0, remainder 0 -- nothing to special-case)
    const unsigned iq = ((n >> 25) + 1u) >> 1;           // rint(64 xi)
    const int drem = (int)(n - (iq << 26));              // (64 xi - iq) 2^26, exact
    const double a = mul_k((double)drem, 1.4629180792671596e-09);      // 2 pi / 2^32; |a| <= pi / 64
    const double z = a * a;
    double ps = -1.9841269841269841270e-04;
    ps = fma_k(ps, z, 8.3333333333333333333e-03);
    ps = fma_k(ps, z, -1.6666666666666666667e-01);
    const double sd = __builtin_fma(a * z, ps, a);
    double pc = 2.4801587301587301587e-05;
    pc = fma_k(pc, z, -1.3888888888888888889e-03);
    pc = fma_k(pc, z, 4.1666666666666666667e-02);
    pc = __builtin_fma(pc, z, -0.5);
    const double cd = __builtin_fma(pc, z, 1.0);
    const double2 Ti = *reinterpret_cast<const double2*>(T + 2 * (kLnTabEntries + (int)(iq & (unsigned)(kScTabEntries - 1))));
    *sn = __builtin_fma(Ti.x, cd, Ti.y * sd);
    *cs = __builtin_fma(Ti.y, cd, -(Ti.x * sd));
}
template <typename R> struct Mx;
template <> struct Mx<double> {
    static LT_DEV double log(double x) { return ::log(x); }
    static LT_DEV double sqrt(double x) { return ::sqrt(x); }
    static LT_DEV double abs(double x) { return ::fabs(x); }
    static LT_DEV double clamp_unit(double c) { return __builtin_fmax(__builtin_fmin(c, 1.0), -1.0); }
    static LT_DEV double sin(double x) { return ::sin(x); }
    static LT_DEV double cos(double x) { return ::cos(x); }
    // the hot-loop forms (restricted ranges, see above)
    static LT_DEV double neg_log(double xi, const double* T) { return neg_log_tab(xi, T); }
    static LT_DEV double sqrt_unit(double x) { return sqrt01(x); }
    static LT_DEV double max0(double x) { return __builtin_fmax(x, 0.0); }
    static LT_DEV unsigned sign_word(double x) { return (unsigned)__double2hiint(x); }      // the word that carries the sign bit
    static LT_DEV double max_tiny(double x) { return __builtin_fmax(x, 1e-300); }
    static LT_DEV double sqrt_pos(double x) { return ltk::sqrt_pos(x); }
    static LT_DEV double rsqrt_pos(double x) { return ltk::rsqrt_pos(x); }
    static LT_DEV double quot(double a, double b) { return fast_div(a, b); }
    static LT_DEV double rcp48(double d) { double r = __builtin_amdgcn_rcp(d); return __builtin_fma(__builtin_fma(-d, r, 1.0), r, r); }      // 1 / d to ~48 bits
    static LT_DEV void sincos_turn(double xi, const double* T, double* s, double* c) { sincos_turn_tab(xi, T, s, c); }
    static LT_DEV double inf() { return __builtin_huge_val(); }
    // rocrand_uniform_double (rocrand_uniform.h:102-109, 454-460): two draws, 53 bits, (0, 1]
    static LT_DEV double uniform(rocrand_state_xorwow* st)
    {
        const unsigned v1 = rocrand(st), v2 = rocrand(st);
        return fma_kk((double)(((unsigned long long)(v2 >> 11) << 32) | v1), 1.1102230246251565404e-16);
    }
    // 32-bit-resolution uniform in (0,1] from ONE draw (rocrand_uniform.h:97-100); used for the three
    // decision/angle uniforms of a step, where 2^-32 is far below any physical resolution
    static LT_DEV double uniform32(rocrand_state_xorwow* st) { return fma_kk((double)rocrand(st), 2.3283064365386962891e-10); }
};
template <> struct Mx<float> {
    static LT_DEV float log(float x) { return ::logf(x); }
    static LT_DEV float sqrt(float x) { return ::sqrtf(x); }
    static LT_DEV float abs(float x) { return ::fabsf(x); }
    static LT_DEV float clamp_unit(float c) { return __builtin_fmaxf(__builtin_fminf(c, 1.0f), -1.0f); }
    static LT_DEV float sin(float x) { return ::sinf(x); }
    static LT_DEV float cos(float x) { return ::cosf(x); }
    static LT_DEV float neg_log(float xi, const double*) { return -::logf(xi); }
    static LT_DEV float sqrt_unit(float x) { return ::sqrtf(__builtin_fmaxf(x, 0.0f)); }
    static LT_DEV float max0(float x) { return __builtin_fmaxf(x, 0.0f); }
    static LT_DEV unsigned sign_word(float x) { return __float_as_uint(x); }
    static LT_DEV float max_tiny(float x) { return __builtin_fmaxf(x, 1e-30f); }
    static LT_DEV float sqrt_pos(float x) { return ::sqrtf(x); }
    static LT_DEV float rsqrt_pos(float x) { return 1.0f / ::sqrtf(x); }
    static LT_DEV float quot(float a, float b) { return a / b; }
    static LT_DEV float rcp48(float d) { return 1.0f / d; }
    static LT_DEV void sincos_turn(float xi, const double*, float* s, float* c) { ::sincospif(2.0f * xi, s, c); }
    static LT_DEV float inf() { return __builtin_huge_valf(); }
    static LT_DEV float uniform(rocrand_state_xorwow* st) { return rocrand_uniform(st); }
    static LT_DEV float uniform32(rocrand_state_xorwow* st) { return rocrand_uniform(st); }
};

template <typename R> LT_DEV R dot3(const R* a, const R* b) { return a[0] * b[0] + a[1] * b[1] + a[2] * b[2]; }
template <typename R> LT_DEV void cross3(const R* a, const R* b, R* o)
{
    o[0] = a[1] * b[2] - a[2] * b[1];
    o[1] = a[2] * b[0] - a[0] * b[2];
    o[2] = a[0] * b[1] - a[1] * b[0];
}
template <typename R> LT_DEV void normalize3(R* v)  // S/vectors.py:6-7
{
    R l = Mx<R>::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    v[0] /= l; v[1] /= l; v[2] /= l;
}

// ---------------------------------------------------------------------------
// geometry queries
// ---------------------------------------------------------------------------
// Moller-Trumbore: triangle_intersect, S/intersects.py:46-104.  NaN == None.
template <typename R>
LT_DEV R tri_hit(const R* o, const R* d, const TriD<R>* T)
{
    const R eps = (R)1e-7;  // :56
    const R nan = (R)__builtin_nanf("");
    R n[3] = {T->n[0], T->n[1], T->n[2]};
    R ddn = dot3(d, n);
    if (Mx<R>::abs(ddn) <= eps) return nan;  // :69-72
    R ab[3] = {T->ab[0], T->ab[1], T->ab[2]};
    R ac[3] = {T->ac[0], T->ac[1], T->ac[2]};
    R pvec[3], qvec[3], tvec[3];
    cross3(d, ac, pvec);              // :74
    R det = dot3(ab, pvec);           // :77
    if (-eps < det && det < eps) return nan;  // :79
    R inv_det = (R)1 / det;
    tvec[0] = o[0] - T->a[0]; tvec[1] = o[1] - T->a[1]; tvec[2] = o[2] - T->a[2];
    R u = dot3(tvec, pvec) * inv_det;  // :86
    if (u < 0 || u > 1) return nan;
    cross3(tvec, ab, qvec);            // :91
    R v = dot3(d, qvec) * inv_det;     // :94
    if (v < 0 || u + v > 1) return nan;
    R t = dot3(ac, qvec) * inv_det;    // :99
    return (t > eps) ? t : nan;        // :101-104
}

// 1 + 2*gamma(3) with float32 machine epsilon, S/intersects.py:229-235
template <typename R> LT_DEV R box_widen()
{
    const R eps = (R)5.9604644775390625e-08;
    return (R)1 + (R)2 * (((R)3 * eps) / ((R)1 - (R)3 * eps));
}

// intersect_bounds, S/intersects.py:179-196
template <typename R>
LT_DEV bool box_hit(const R* lo, const R* hi, const R* o, const R* inv_d, R tmax)
{
    R t0 = 0, t1 = tmax;
    const R widen = box_widen<R>();
#pragma unroll
    for (int i = 0; i < 3; i++) {
        R tn = (lo[i] - o[i]) * inv_d[i];
        R tf = (hi[i] - o[i]) * inv_d[i];
        if (tn > tf) { R s = tn; tn = tf; tf = s; }
        tf *= widen;
        t0 = tn > t0 ? tn : t0;
        t1 = tf < t1 ? tf : t1;
        if (t0 > t1) return false;
    }
    return true;
}

// nearest-hit bookkeeping: predicate EPSILON < t < min_distance
// (S/bvh_new.py:438); equal t -> lower primitive index (what a brute-force
// scan in index order returns; the reference traversal's bug B3 is not kept).
template <typename R>
LT_DEV void consider(const TriD<R>* tris, int i, const R* o, const R* d, int& bi, R& bt)
{
    R t = tri_hit(o, d, &tris[i]);
    if (t == t && t > (R)1e-6) {
        if (t < bt || (t == bt && bi >= 0 && i < bi)) { bt = t; bi = i; }
    }
}

// Stackless pre-order walk: nodes are stored in DFS order and every node carries the index of the first
// node AFTER its subtree (NodeD::skip, filled on the host), so a missed box or a finished leaf jumps there and
// a hit interior node simply continues with cur + 1.  No per-lane stack, no VGPR-indexed arrays; the result
// (nearest hit, ties to the lower primitive index) does not depend on the visiting order.
template <typename R>
LT_DEV void nearest_bvh(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const R* o,
                        const R* d, R tmax, int& prim, R& t_out)
{
    int bi = -1; R bt = tmax;
    const R inv_d[3] = {(R)1 / d[0], (R)1 / d[1], (R)1 / d[2]};  // S/bvh_new.py:418
    const R slack = box_widen<R>();
    int cur = 0;
    while (cur < n_nodes) {
        const NodeD<R>* nd = &nodes[cur];
        const R lo[3] = {nd->lo[0], nd->lo[1], nd->lo[2]};
        const R hi[3] = {nd->hi[0], nd->hi[1], nd->hi[2]};
        const int np = nd->n_prims, off = nd->offset, skip = nd->skip;
        if (box_hit(lo, hi, o, inv_d, bt * slack)) {
            if (np > 0) {
                for (int k = 0; k < np; k++) consider(tris, off + k, o, d, bi, bt);
                cur = skip;
            } else {
                cur = cur + 1;
            }
        } else {
            cur = skip;
        }
    }
    prim = bi; t_out = bi >= 0 ? bt : Mx<R>::inf();
}

// The same search in FRONT-TO-BACK order -- the reference enters the child on the ray's side of the split plane first
// (S/bvh_new.py:455-458: `if dir_is_neg[node.axis]` the second child, else the first) so that the nearest hit found early
// prunes the far boxes -- still without a stack: for each of the 8 sign patterns of a direction the host threads the tree
// once (lt_api.cpp: bvh_octant_links): first[i] = the child of interior node i the ray meets first, after[i] = where the
// search continues once the subtree of i is done (its sibling, or its parent's continuation; n_nodes = finished).
// `links` points at this ray's pattern: first[0 .. n), after[0 .. n).  The result does not depend on the order (nearest
// hit, ties to the lower primitive index), only the number of boxes and triangles tested does.
// ANY: stop at the first accepted hit (shadow rays: is anything nearer than tmax?).
template <typename R>
LT_DEV void nearest_bvh_ordered(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const int16_t* links, const R* o,
                                const R* d, R tmax, int& prim, R& t_out, bool ANY = false)
{
    int bi = -1; R bt = tmax;
    const R inv_d[3] = {(R)1 / d[0], (R)1 / d[1], (R)1 / d[2]};  // S/bvh_new.py:418
    const R slack = box_widen<R>();
    const int16_t* first = links; const int16_t* after = links + n_nodes;
    int cur = 0;
    while (cur < n_nodes) {
        const NodeD<R>* nd = &nodes[cur];
        const R lo[3] = {nd->lo[0], nd->lo[1], nd->lo[2]};
        const R hi[3] = {nd->hi[0], nd->hi[1], nd->hi[2]};
        const int np = nd->n_prims, off = nd->offset;
        if (box_hit(lo, hi, o, inv_d, bt * slack)) {
            if (np > 0) {
                for (int k = 0; k < np; k++) consider(tris, off + k, o, d, bi, bt);
                if (ANY && bi >= 0) break;
                cur = after[cur];
            } else {
                cur = first[cur];
            }
        } else {
            cur = after[cur];
        }
    }
    prim = bi; t_out = bi >= 0 ? bt : Mx<R>::inf();
}
LT_DEV int ray_octant(const double* d) { return (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0); }
LT_DEV int ray_octant(const float* d) { return (d[0] < 0 ? 1 : 0) | (d[1] < 0 ? 2 : 0) | (d[2] < 0 ? 4 : 0); }
// the walks' BVH search: front to back where the mesh has link tables (every mesh of <= 32767 nodes), in storage order otherwise
template <typename R>
LT_DEV void nearest_bvh_walk(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const int16_t* links, const R* o,
                             const R* d, R tmax, int& prim, R& t_out)
{
    if (links) nearest_bvh_ordered(tris, nodes, n_nodes, links + ray_octant(d) * 2 * n_nodes, o, d, tmax, prim, t_out);
    else nearest_bvh(tris, nodes, n_nodes, o, d, tmax, prim, t_out);
}

template <typename R>
LT_DEV void nearest_brute(const TriD<R>* tris, int n_tris, const R* o, const R* d, R tmax,
                          int& prim, R& t_out)
{
    int bi = -1; R bt = tmax;
    for (int i = 0; i < n_tris; i++) consider(tris, i, o, d, bi, bt);
    prim = bi; t_out = bi >= 0 ? bt : Mx<R>::inf();
}

// A query whose hop is shorter than the cell's c2 (c4) bound can only hit the 2 (4) triangles listed in the cell's
// clearance record (WalkParams::clear): they are tested directly -- same tri_hit, same nearest / tie rule, hence the
// same answer as the BVH walk, which is left to the hops that reach farther.
template <typename R>
LT_DEV void nearest_listed(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const int16_t* links, const uint4 rec, const R* o,
                           const R* d, R tmax, int& prim, R& t_out)
{
    const float c2 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rec.y & 0xffffu));
    const float c4 = (float)__builtin_bit_cast(_Float16, (unsigned short)(rec.y >> 16));
    if (tmax < (R)c4) {
        int bi = -1; R bt = tmax;
        const int i0 = (int)(rec.z & 0xffffu), i1 = (int)(rec.z >> 16);
        if (i0 != 0xffff) consider(tris, i0, o, d, bi, bt);
        if (i1 != 0xffff) consider(tris, i1, o, d, bi, bt);
        if (!(tmax < (R)c2)) {
            const int i2 = (int)(rec.w & 0xffffu), i3 = (int)(rec.w >> 16);
            if (i2 != 0xffff) consider(tris, i2, o, d, bi, bt);
            if (i3 != 0xffff) consider(tris, i3, o, d, bi, bt);
        }
        prim = bi; t_out = bi >= 0 ? bt : Mx<R>::inf();
    } else {
        nearest_bvh(tris, nodes, n_nodes, o, d, tmax, prim, t_out);
    }
}

// Can a hop of length s from o along d reach the PLANE of triangle T at all?  The hit parameter of tri_hit is, analytically,
// t = ((a - o).n) / (d.n); the hop misses the triangle for certain when t <= 0 (the plane lies behind) or t >= s.  This is
// a REJECTION test in front of tri_hit, never a replacement: it answers "no" only where tri_hit's own t -- computed another
// way, so equal to this one up to rounding -- cannot lie inside (EPSILON, s): rays nearly parallel to the plane, where the
// two parameters may differ by more than the margins, are always passed on.  On the Cornell cavity 89 % of the surface
// queries find nothing -- a photon near a wall whose hop does not reach it -- and nearly all of those stop here.
template <typename R> LT_DEV bool plane_within_reach(const TriD<R>* T, R px, R py, R pz, R ux, R uy, R uz, R s)
{
    const R nx = T->n[0], ny = T->n[1], nz = T->n[2];
    const R ddn = ux * nx + uy * ny + uz * nz;
    const R dist = (T->a[0] - px) * nx + (T->a[1] - py) * ny + (T->a[2] - pz) * nz;      // = t * ddn
    const R addn = Mx<R>::abs(ddn);
    const R graze = sizeof(R) == 8 ? (R)1e-3 : (R)1e-2, behind = sizeof(R) == 8 ? (R)1e-7 : (R)1e-2,
            margin = sizeof(R) == 8 ? (R)(1.0 + 1e-4) : (R)(1.0 + 1e-2);
    if (!(addn > graze)) return true;
    const R tdd = ddn > 0 ? dist : -dist;       // t * |ddn|
    return tdd >= -behind * addn && tdd <= s * addn * margin;
}

// The three decision uniforms a mesh walk holds while its step waits for a surface query: the f64 XORWOW walk keeps the raw
// 32-bit draws (3 VGPRs instead of 6) and converts them when the step resumes -- the same conversion, the same value.
template <typename R, bool TABLE> struct HeldU {
    typedef R type;
    static LT_DEV R get(R v) { return v; }
    static LT_DEV R draw(rocrand_state_xorwow* st) { return Mx<R>::uniform32(st); }
    static LT_DEV R as_real(R v) { return v; }
    static LT_DEV R from_real(R v) { return v; }
    static LT_DEV R neg_log(R v, const double* T) { return Mx<R>::neg_log(v, T); }
    static LT_DEV void sincos_turn(R v, const double* T, R* s, R* c) { Mx<R>::sincos_turn(v, T, s, c); }
};
template <> struct HeldU<double, false> {
    typedef unsigned type;
    static LT_DEV double get(unsigned v) { return fma_kk((double)v, 2.3283064365386962891e-10); }     // Mx<double>::uniform32
    static LT_DEV unsigned draw(rocrand_state_xorwow* st) { return rocrand(st); }
    static LT_DEV double as_real(unsigned v) { return (double)v; }       // (exact both ways: the march kernel parks it in an LDS array of R)
    static LT_DEV unsigned from_real(double v) { return (unsigned)v; }
    static LT_DEV double neg_log(unsigned v, const double* T) { return neg_log_raw(v, T); }
    static LT_DEV void sincos_turn(unsigned v, const double* T, double* s, double* c) { sincos_turn_raw(v, T, s, c); }
};

// c0 of the cell that holds (px, py, pz): -1 outside the grid (always query)
template <typename R> LT_DEV float march_clearance(const MarchGrid& G, R px, R py, R pz)
{
    const R cx = (px - LT_GV(org, 0)) * LT_GV(inv, 0), cy = (py - LT_GV(org, 1)) * LT_GV(inv, 1), cz = (pz - LT_GV(org, 2)) * LT_GV(inv, 2);
    if (cx >= 0 && cx < LT_GV(fn, 0) && cy >= 0 && cy < LT_GV(fn, 1) && cz >= 0 && cz < LT_GV(fn, 2))
        return __uint_as_float(G.cell[((size_t)(int)cz * G.ny + (int)cy) * G.nx + (int)cx].x & ~kMarchCountMask);
    return -1.0f;
}

// Grid march (GEOM 2, MarchGrid in lt_internal.hpp): the hop segment o + t d, t in [t0, tmax), is walked cell by cell,
// front to back -- the order the reference's traversal aims for by visiting the near child first (S/bvh_new.py:455-458).
// Every visited cell's candidates are tested with consider() (= tri_hit + the nearest / tie rule), so the result equals
// a brute-force scan:  a triangle hit at parameter t lies, up to rounding, in the cell the march is in at t, and the lists
// hold every triangle within a margin (1e-5 of a cell, >= 1e4 x any rounding here) of the cell;  the march stops as soon as
// the best hit so far is not beyond the exit of the current cell (everything nearer has been tested) or the segment ends.
// Free space is crossed c0 at a time: no triangle lies within c0 of ANY point of a cell, the exit point included.
// t0: a distance from o already known to be clear (the c0 of o's cell; 0 = none).
// Origins outside the grid (open meshes) take the BVH.
template <typename R>
LT_DEV void nearest_march(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const MarchGrid& G, const R* o,
                          const R* d, R tmax, R t0, int& prim, R& t_out)
{
    const R inf = Mx<R>::inf();
    const R gx = LT_GV(org, 0), gy = LT_GV(org, 1), gz = LT_GV(org, 2);
    const R ix_ = LT_GV(inv, 0), iy_ = LT_GV(inv, 1), iz_ = LT_GV(inv, 2);
    const R hx = LT_GV(h, 0), hy = LT_GV(h, 1), hz = LT_GV(h, 2);
    const R fnx = LT_GV(fn, 0), fny = LT_GV(fn, 1), fnz = LT_GV(fn, 2);
    const R tol = (R)1e-4;                       // in cells: positions this close outside the grid are clamped into it
    {
        const R fx = (o[0] - gx) * ix_, fy = (o[1] - gy) * iy_, fz = (o[2] - gz) * iz_;
        if (!(fx >= -tol && fx <= fnx + tol && fy >= -tol && fy <= fny + tol && fz >= -tol && fz <= fnz + tol)) {
            nearest_bvh(tris, nodes, n_nodes, o, d, tmax, prim, t_out);
            return;
        }
    }
    int bi = -1; R bt = tmax;
    // an axis the ray does not move along never steps (1 / d would be +-inf, 0 * inf a NaN)
    const R tiny = sizeof(R) == 8 ? (R)1e-200 : (R)1e-30;
    const bool mx = Mx<R>::abs(d[0]) > tiny, my = Mx<R>::abs(d[1]) > tiny, mz = Mx<R>::abs(d[2]) > tiny;
    const R idx_ = mx ? (R)1 / d[0] : inf, idy_ = my ? (R)1 / d[1] : inf, idz_ = mz ? (R)1 / d[2] : inf;
    const R tdx = mx ? hx * Mx<R>::abs(idx_) : inf, tdy = my ? hy * Mx<R>::abs(idy_) : inf, tdz = mz ? hz * Mx<R>::abs(idz_) : inf;
    const int sx = d[0] > 0 ? 1 : -1, sy = d[1] > 0 ? 1 : -1, sz = d[2] > 0 ? 1 : -1;
    int cx = 0, cy = 0, cz = 0;
    R tmx = inf, tmy = inf, tmz = inf;
    R t = t0 > 0 ? t0 : (R)0;
    bool inside = true;
    // (re)start the DDA at parameter t: cell of the point, parameter of the next cell wall on every axis
    auto restart = [&]() {
        const R fx = (o[0] + t * d[0] - gx) * ix_, fy = (o[1] + t * d[1] - gy) * iy_, fz = (o[2] + t * d[2] - gz) * iz_;
        inside = fx >= -tol && fx <= fnx + tol && fy >= -tol && fy <= fny + tol && fz >= -tol && fz <= fnz + tol;
        cx = (int)__builtin_floor(fx); cy = (int)__builtin_floor(fy); cz = (int)__builtin_floor(fz);
        cx = cx < 0 ? 0 : (cx >= G.nx ? G.nx - 1 : cx);
        cy = cy < 0 ? 0 : (cy >= G.ny ? G.ny - 1 : cy);
        cz = cz < 0 ? 0 : (cz >= G.nz ? G.nz - 1 : cz);
        tmx = mx ? (gx + (R)(cx + (sx > 0 ? 1 : 0)) * hx - o[0]) * idx_ : inf;
        tmy = my ? (gy + (R)(cy + (sy > 0 ? 1 : 0)) * hy - o[1]) * idy_ : inf;
        tmz = mz ? (gz + (R)(cz + (sz > 0 ? 1 : 0)) * hz - o[2]) * idz_ : inf;
    };
    if (!(t < bt)) { prim = -1; t_out = inf; return; }
    restart();
    const R jump_min = (R)1.5 * (hx < hy ? (hx < hz ? hx : hz) : (hy < hz ? hy : hz));   // a jump must beat a DDA step
    while (inside) {
        const uint2 rec = G.cell[((size_t)cz * G.ny + cy) * G.nx + cx];
        const R tex = tmx < tmy ? (tmx < tmz ? tmx : tmz) : (tmy < tmz ? tmy : tmz);
        const unsigned n = rec.x & kMarchCountMask;
        if (n == kMarchCountMask) { nearest_bvh(tris, nodes, n_nodes, o, d, tmax, prim, t_out); return; }   // a cell with a long list
        const uint32_t* L = G.list + rec.y;
        for (unsigned k = 0; k < n; k++) consider(tris, (int)L[k], o, d, bi, bt);
        if (!(tex < bt)) break;          // the segment (or everything nearer than the best hit) ends inside this cell
        const R clr = (R)__uint_as_float(rec.x & ~kMarchCountMask);
        if (clr > jump_min) {
            t = tex + clr;
            if (!(t < bt)) break;
            restart();
        } else if (tmx <= tmy && tmx <= tmz) { cx += sx; tmx += tdx; inside = cx >= 0 && cx < G.nx; }
        else if (tmy <= tmz) { cy += sy; tmy += tdy; inside = cy >= 0 && cy < G.ny; }
        else { cz += sz; tmz += tdz; inside = cz >= 0 && cz < G.nz; }
    }
    prim = bi; t_out = bi >= 0 ? bt : inf;
}

// The walk's form of the grid march: the lanes of a wave that need a surface query are served TOGETHER.
// Marching lane by lane (nearest_march) is a chain of dependent loads per lane -- cell record, candidate id, triangle --
// and the wave pays for its longest lane: instrumented on the reference's teapot, the average lane tested 2.4 triangles,
// the longest 19, at ~3000 cycles per link of the chain (96 000 cycles per service, 67 % of the walk).  Here the lanes only
// march (march_round: one cell per lane per round, candidates pushed as (lane, list position) items into a queue in LDS)
// and the WHOLE wave tests the queued (ray, triangle) pairs, one pair per lane per round, whoever's they are
// (march_drain): the triangle tests run on full waves with independent loads, and the nearest hit of every ray is folded
// with LDS atomics -- min over the bit patterns of t (positive floats order like unsigned integers), then min over the
// ids of the triangles that reach that t: the nearest / tie rule of consider().  A ray's candidates are the triangles of
// the cells its segment crosses up to the best hit known when a cell is entered; tests beyond the nearest hit cannot
// change it, so the answer is the brute-force scan's.
// The march is position based: a round computes the cell of the point at parameter tcur, and tcur then moves to the
// cell's exit plus a nudge (far above rounding, far below the margin the candidate lists were built with, so whatever
// the nudge skips is listed in the cell just left) -- or past the exit by the cell's clearance.  Only tcur lives in a
// register between rounds, which lets the walk run rounds as part of its own iterations (lt_walk_kernel.inc, batch 2).
constexpr unsigned kMarchQ = 256;               // queue items per wave: four rounds of march_drain
constexpr unsigned kMarchNone = 0xffffffffu;
template <typename R> struct MarchBits { typedef unsigned long long type; };
template <> struct MarchBits<float> { typedef unsigned type; };
template <typename R> struct MarchWave {        // per wave, in LDS
    typename MarchBits<R>::type slot_t[64];     // best t per ray (bits), starts at the ray's tmax
    unsigned slot_i[64];                        // triangle of that t (lowest id on ties)
    R ray[9][64];                               // origin, direction, 1 / direction (inf for an axis the ray does not move along)
    R held[4][64];                              // walk_kernel_m: the prepared step of a lane in a query (free path, three decision uniforms)
    unsigned q_item[kMarchQ];                   // lane << 26 | position in MarchGrid::list
    typename MarchBits<R>::type q_t[kMarchQ];   // t of the item's test (all ones: no hit)
};
LT_DEV unsigned long long march_bits(double t) { return (unsigned long long)__double_as_longlong(t); }
LT_DEV unsigned march_bits(float t) { return __float_as_uint(t); }
LT_DEV double march_unbits(unsigned long long b) { return __longlong_as_double((long long)b); }
LT_DEV float march_unbits(unsigned b) { return __uint_as_float(b); }

// a lane's query enters the march: ray and best-hit slot into LDS.  false: the origin lies outside the grid (open meshes)
// -- that query takes the BVH.
template <typename R>
LT_DEV bool march_enter(const MarchGrid& G, MarchWave<R>* W, R px, R py, R pz, R ux, R uy, R uz, R tmax)
{
    const unsigned lane = threadIdx.x & 63u;
    const R inf = Mx<R>::inf(), tiny = sizeof(R) == 8 ? (R)1e-200 : (R)1e-30, tol = (R)1e-4;
    W->ray[0][lane] = px; W->ray[1][lane] = py; W->ray[2][lane] = pz;
    W->ray[3][lane] = ux; W->ray[4][lane] = uy; W->ray[5][lane] = uz;
    W->ray[6][lane] = Mx<R>::abs(ux) > tiny ? (R)1 / ux : inf;
    W->ray[7][lane] = Mx<R>::abs(uy) > tiny ? (R)1 / uy : inf;
    W->ray[8][lane] = Mx<R>::abs(uz) > tiny ? (R)1 / uz : inf;
    W->slot_t[lane] = march_bits(tmax); W->slot_i[lane] = kMarchNone;
    const R fx = (px - LT_GV(org, 0)) * LT_GV(inv, 0), fy = (py - LT_GV(org, 1)) * LT_GV(inv, 1), fz = (pz - LT_GV(org, 2)) * LT_GV(inv, 2);
    return fx >= -tol && fx <= LT_GV(fn, 0) + tol && fy >= -tol && fy <= LT_GV(fn, 1) + tol && fz >= -tol && fz <= LT_GV(fn, 2) + tol;
}

// One round: every marching lane visits the cell of its point at tcur, queues the cell's candidates and moves tcur on.
// marching goes false when the segment (or everything nearer than the best hit) ends, or the ray leaves the grid;
// heavy: the cell has a long list (that query takes the BVH); dirty: the lane has items in the queue; qn: items in the
// queue (wave-uniform); returns true when some lane could not queue its candidates (it stays on its cell: drain, then go on).
template <typename R>
LT_DEV bool march_round(const MarchGrid& G, MarchWave<R>* W, bool& marching, bool& heavy, bool& dirty, R& tcur, unsigned& qn)
{
    const unsigned lane = threadIdx.x & 63u;
    const R inf = Mx<R>::inf(), tol = (R)1e-4;
    const R hx = LT_GV(h, 0), hy = LT_GV(h, 1), hz = LT_GV(h, 2);
    const R nudge = sizeof(R) == 8 ? (R)G.nudge64 : (R)G.nudge32;
    unsigned n = 0, lst = 0;
    R tex = inf, bt = inf; float clr = 0.0f;
    bool m = marching;
    if (m) {
        bt = march_unbits(W->slot_t[lane]);
        if (!(tcur < bt)) m = false;
    }
    if (m) {
        const R px = W->ray[0][lane], py = W->ray[1][lane], pz = W->ray[2][lane];
        const R ux = W->ray[3][lane], uy = W->ray[4][lane], uz = W->ray[5][lane];
        const R fx = (px + tcur * ux - LT_GV(org, 0)) * LT_GV(inv, 0), fy = (py + tcur * uy - LT_GV(org, 1)) * LT_GV(inv, 1),
                fz = (pz + tcur * uz - LT_GV(org, 2)) * LT_GV(inv, 2);
        // out of the grid, or on its outer wall and heading out: nothing beyond the root bounds (the walls themselves are
        // listed in the outermost cells, which the march has visited by then)
        if (!(fx >= -tol && fx <= LT_GV(fn, 0) + tol && fy >= -tol && fy <= LT_GV(fn, 1) + tol && fz >= -tol && fz <= LT_GV(fn, 2) + tol) ||
            (ux > 0 ? fx >= LT_GV(fn, 0) : (ux < 0 && fx <= 0)) || (uy > 0 ? fy >= LT_GV(fn, 1) : (uy < 0 && fy <= 0)) ||
            (uz > 0 ? fz >= LT_GV(fn, 2) : (uz < 0 && fz <= 0))) m = false;
        else {
            int cx = (int)__builtin_floor(fx), cy = (int)__builtin_floor(fy), cz = (int)__builtin_floor(fz);
            cx = cx < 0 ? 0 : (cx >= G.nx ? G.nx - 1 : cx);
            cy = cy < 0 ? 0 : (cy >= G.ny ? G.ny - 1 : cy);
            cz = cz < 0 ? 0 : (cz >= G.nz ? G.nz - 1 : cz);
            const uint2 rec = G.cell[((size_t)cz * G.ny + cy) * G.nx + cx];
            n = rec.x & kMarchCountMask; lst = rec.y; clr = __uint_as_float(rec.x & ~kMarchCountMask);
            if (n == kMarchCountMask) { heavy = true; m = false; n = 0; }
            // parameter at which the ray leaves this cell (inf on an axis it does not move along: 1 / d is inf there and the
            // numerator is never 0 * inf because it is taken as inf outright)
            const R ix = W->ray[6][lane], iy = W->ray[7][lane], iz = W->ray[8][lane];
            const R ex = ix < inf && ix > -inf ? (LT_GV(org, 0) + (R)(cx + (ux > 0 ? 1 : 0)) * hx - px) * ix : inf;
            const R ey = iy < inf && iy > -inf ? (LT_GV(org, 1) + (R)(cy + (uy > 0 ? 1 : 0)) * hy - py) * iy : inf;
            const R ez = iz < inf && iz > -inf ? (LT_GV(org, 2) + (R)(cz + (uz > 0 ? 1 : 0)) * hz - pz) * iz : inf;
            tex = ex < ey ? (ex < ez ? ex : ez) : (ey < ez ? ey : ez);
        }
    }
    if (!m) marching = false;
    // exclusive prefix sum of n (< 64) over the lanes, bit by bit with ballots: scalar / VALU work only.  (Six __shfl_up
    // steps -- ds_bpermute round trips -- cost the walk 3.5-4 %: teapot 18.6 -> 19.3-19.4e9 steps/s, profiles/r03a_*.log)
    unsigned excl = 0, total_ = 0;
#pragma unroll
    for (unsigned b = 0; b < 6; b++) {
        const unsigned long long mb = __ballot((n >> b) & 1u);
        excl += __builtin_amdgcn_mbcnt_hi((unsigned)(mb >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)mb, 0u)) << b;
        total_ += (unsigned)__popcll(mb) << b;
    }
    const unsigned incl = excl + n, cap = kMarchQ - qn;
    const bool blocked = incl > cap;                            // does not fit: the lane stays on this cell until the queue has drained
    const unsigned long long bm = __ballot(blocked);            // (the blocked lanes are a suffix: the prefix sums ascend)
    const unsigned pushed = bm ? (unsigned)__builtin_amdgcn_readlane((int)excl, __ffsll((long long)bm) - 1) : total_;
    if (!blocked) for (unsigned k = 0; k < n; k++) W->q_item[qn + excl + k] = (lane << 26) | (lst + k);
    qn += pushed;
    if (m && !blocked) {
        if (n) dirty = true;
        const R jump_min = (R)1.5 * (hx < hy ? (hx < hz ? hx : hz) : (hy < hz ? hy : hz));
        if (!(tex < bt)) marching = false;          // the segment (or everything nearer than the best hit) ends in this cell
        else {
            // free space: no triangle lies within clr of ANY point of this cell, its exit point included
            R tn = (R)clr > jump_min ? tex + (R)clr : tex + nudge;
            const R floor_ = tcur + nudge;          // (a round always moves on, whatever rounding did to tex)
            tcur = tn > floor_ ? tn : floor_;
            if (!(tcur < bt)) marching = false;
        }
    }
    return bm != 0ull;
}

// The wave tests the queued pairs, one per lane per round, and folds the hits into the rays' slots.
template <typename R>
LT_DEV void march_drain(const TriD<R>* tris, const MarchGrid& G, MarchWave<R>* W, unsigned& qn)
{
    typedef typename MarchBits<R>::type UB;
    const unsigned lane = threadIdx.x & 63u;
    __builtin_amdgcn_fence(__ATOMIC_RELEASE, "wavefront");
    for (unsigned j0 = 0; j0 < qn; j0 += 64u) {
        const unsigned j = j0 + lane;
        UB tb = ~(UB)0;
        if (j < qn) {
            const unsigned item = W->q_item[j], src = item >> 26;
            const int id = (int)G.list[item & 0x3ffffffu];
            const R o[3] = {W->ray[0][src], W->ray[1][src], W->ray[2][src]};
            const R d[3] = {W->ray[3][src], W->ray[4][src], W->ray[5][src]};
            const R t = tri_hit(o, d, &tris[id]);
            if (t == t && t > (R)1e-6) {        // consider(): EPSILON < t; "< best" is the atomic min itself
                tb = march_bits(t);
                const UB old = __hip_atomic_fetch_min(&W->slot_t[src], tb, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
                if (tb < old) W->slot_i[src] = kMarchNone;      // a nearer hit: the ids collected for the old t are void
            }
        }
        W->q_t[j] = tb;
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    for (unsigned j0 = 0; j0 < qn; j0 += 64u) {
        const unsigned j = j0 + lane;
        if (j < qn) {
            const UB tb = W->q_t[j];
            const unsigned item = W->q_item[j], src = item >> 26;
            if (tb != ~(UB)0 && tb == W->slot_t[src])
                (void)__hip_atomic_fetch_min(&W->slot_i[src], G.list[item & 0x3ffffffu], __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_WORKGROUP);
        }
    }
    __builtin_amdgcn_fence(__ATOMIC_ACQ_REL, "wavefront");
    qn = 0;
}

// a finished query's answer (the lane's slot, or the BVH for the queries the grid could not take)
template <typename R>
LT_DEV void march_result(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const int16_t* links, MarchWave<R>* W, bool use_bvh, R px, R py, R pz,
                         R ux, R uy, R uz, R tmax, int& prim, R& t_out)
{
    const unsigned lane = threadIdx.x & 63u;
    prim = -1; t_out = Mx<R>::inf();
    if (use_bvh) {
        const R o[3] = {px, py, pz}, d[3] = {ux, uy, uz};
        nearest_bvh_walk(tris, nodes, n_nodes, links, o, d, tmax, prim, t_out);
    } else {
        const typename MarchBits<R>::type b = W->slot_t[lane];
        if (b < march_bits(tmax)) { prim = (int)W->slot_i[lane]; t_out = march_unbits(b); }
    }
}

// all of it for one batch of rays (one per lane; want: this lane has a ray): rounds until every ray is answered
template <typename R>
LT_DEV void march_service(const TriD<R>* tris, const NodeD<R>* nodes, int n_nodes, const MarchGrid& G, MarchWave<R>* W,
                          bool want, R px, R py, R pz, R ux, R uy, R uz, R tmax, R t0, int& prim, R& t_out)
{
    bool marching = false, heavy = false, dirty = false, outside = false;
    R tcur = t0 > 0 ? t0 : (R)0;
    if (want) { outside = !march_enter(G, W, px, py, pz, ux, uy, uz, tmax); marching = !outside; }
    unsigned qn = 0;
    while (__any(marching) || qn > 0) {
        bool full = false;
        if (__any(marching)) full = march_round(G, W, marching, heavy, dirty, tcur, qn);
        if (qn > 0 && (full || qn >= 64u || !__any(marching))) march_drain(tris, G, W, qn);
    }
    if (want) march_result(tris, nodes, n_nodes, (const int16_t*)nullptr, W, outside || heavy, px, py, pz, ux, uy, uz, tmax, prim, t_out);
    else { prim = -1; t_out = Mx<R>::inf(); }
}

// ---------------------------------------------------------------------------
// sampling
// ---------------------------------------------------------------------------
// henyey_greenstein, S/medium_samples.py:14-16
template <typename R> LT_DEV R hg_pdf(R cos_t, R g)
{
    R denom = (R)1 + g * g + (R)2 * g * cos_t;
    return (R)0.07957747154594767 * ((R)1 - g * g) / (denom * Mx<R>::sqrt(denom));
}

// HG inverse CDF, deflection-angle convention
template <typename R> LT_DEV R hg_sample(R xi, R g, R one_m_g2, R one_p_g2, R inv_2g)
{
    R c;
    if (g == 0) c = (R)2 * xi - (R)1;
    else {
        R t = Mx<R>::quot(one_m_g2, (R)1 - g + (R)2 * g * xi);
        c = (one_p_g2 - t * t) * inv_2g;
    }
    return Mx<R>::clamp_unit(c);   // [-1, 1]
}

// ... as the walk takes it: the denominator from the medium record ((1 - g) + (2 g) xi, one fused product) and NO clamp -- the
// value leaves [-1, 1] by a rounding at most, which sqrt01 (sin theta) absorbs and the direction update does not notice.
template <typename R> LT_DEV R hg_sample_walk(R xi, const MedD<R>* M)
{
    if (M->g == 0) return (R)2 * xi - (R)1;
    const R t = Mx<R>::quot(M->one_m_g2, M->two_g * xi + M->one_m_g);
    return (M->one_p_g2 - t * t) * M->inv_2g;
}

// create_orthonormal_system, S/utils.py:72-80
template <typename R> LT_DEV void onb(const R* n, R* v2, R* v3)
{
    if (Mx<R>::abs(n[0]) > Mx<R>::abs(n[1])) {
        R l = Mx<R>::sqrt(n[0] * n[0] + n[2] * n[2]);
        v2[0] = -n[2] / l; v2[1] = 0; v2[2] = n[0] / l;
    } else {
        R l = Mx<R>::sqrt(n[1] * n[1] + n[2] * n[2]);
        v2[0] = 0; v2[1] = n[2] / l; v2[2] = -n[1] / l;
    }
    cross3(n, v2, v3);
}

// concentric_sample_disk, S/utils.py:115-128
// (LEAN: the surface renderers' build -- sin / cos by sincos_small_f64 instead of the general library routines)
template <typename R, bool LEAN = false> LT_DEV void disk(R u0, R u1, R* d)
{
    R ox = (R)2 * u0 - (R)1, oy = (R)2 * u1 - (R)1;
    if (ox == 0 && oy == 0) { d[0] = 0; d[1] = 0; return; }
    R r, theta;
    if (Mx<R>::abs(ox) > Mx<R>::abs(oy)) { r = ox; theta = (R)0.7853981633974483 * (oy / ox); }
    else { r = oy; theta = (R)1.5707963267948966 - (R)0.7853981633974483 * (ox / oy); }
    if constexpr (LEAN && sizeof(R) == 8) {
        double sn, cs;
        sincos_small_f64(theta, &sn, &cs);
        d[0] = r * cs; d[1] = r * sn;
    } else {
        d[0] = r * Mx<R>::cos(theta); d[1] = r * Mx<R>::sin(theta);
    }
}

// cosine_weighted_hemisphere_sampling, S/utils.py:132-161
// (cosine_hemi_frame: the same with the frame of n handed in -- the walk's source normal is the same for every photon, its
// frame is made once per workgroup and kept in LDS instead of being carried through the hot loop in registers)
template <typename R> LT_DEV void cosine_hemi_frame(const R* n, const R* v2, const R* v3, const R* wi_in, R u0, R u1, R* out)
{
    R wiz = -wi_in[2];  // :133
    R d[2]; disk(u0, u1, d);
    R zz = (R)1 - d[0] * d[0] - d[1] * d[1];
    R z = Mx<R>::sqrt(zz > 0 ? zz : (R)0);  // :138
    R oz = z;
    if (wiz < 0) oz = -oz;  // :145-146
    R pdf = (wiz * oz > 0) ? Mx<R>::abs(z) * (R)0.3183098861837907 : (R)0;  // :149-152
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = d[0] * v2[k] + d[1] * v3[k] + oz * n[k];  // :154-157
    out[3] = pdf;
}
template <typename R, bool LEAN = false> LT_DEV void cosine_hemi(const R* n, const R* wi_in, R u0, R u1, R* out)
{
    R wiz = -wi_in[2];  // :133
    R d[2]; disk<R, LEAN>(u0, u1, d);
    R zz = (R)1 - d[0] * d[0] - d[1] * d[1];
    R z = Mx<R>::sqrt(zz > 0 ? zz : (R)0);  // :138
    R oz = z;
    R v2[3], v3[3]; onb(n, v2, v3);
    if (wiz < 0) oz = -oz;  // :145-146
    R pdf = (wiz * oz > 0) ? Mx<R>::abs(z) * (R)0.3183098861837907 : (R)0;  // :149-152
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = d[0] * v2[k] + d[1] * v3[k] + oz * n[k];  // :154-157
    out[3] = pdf;
}

// get_reflected_direction, S/brdf.py:8-9
template <typename R> LT_DEV void reflect(const R* v, const R* n, R* o)
{
    R k = (R)2 * dot3(v, n);
    o[0] = v[0] - k * n[0]; o[1] = v[1] - k * n[1]; o[2] = v[2] - k * n[2];
    normalize3(o);
}

// the walk's form: same expression, normalised with one reciprocal square root (|o| = 1 to rounding already)
template <typename R> LT_DEV void reflect_walk(const R* v, const R* n, R* o)
{
    R k = (R)2 * dot3(v, n);
    o[0] = v[0] - k * n[0]; o[1] = v[1] - k * n[1]; o[2] = v[2] - k * n[2];
    const R inv = Mx<R>::rsqrt_pos(o[0] * o[0] + o[1] * o[1] + o[2] * o[2]);
    o[0] *= inv; o[1] *= inv; o[2] *= inv;
}

// Dielectric boundary (App. C.5): exact unpolarised Fresnel + the refraction
// vector of S/path_tracing_fix1.py:107-114 with Nr = n1/n2; TIR when the
// radicand <= 0 (:110).  nf faces the incoming photon.
template <typename R> LT_DEV R boundary(const R* d, const R* nf, R n1, R n2, R* cos_t_out, R* refr)
{
    R cos_i = -dot3(d, nf);
    if (n1 == n2) {
        *cos_t_out = cos_i; refr[0] = d[0]; refr[1] = d[1]; refr[2] = d[2];
        return 0;
    }
    // lean arithmetic (Mx::quot / sqrt_pos / rsqrt_pos: v_rcp / v_rsq + Newton, <= 1-2 ulp): this block runs whenever
    // ANY lane of the wave sits on an interface, which in a layered medium is most wave-steps
    R Nr = Mx<R>::quot(n1, n2);
    R rad = (R)1 - Nr * Nr * ((R)1 - cos_i * cos_i);
    if (rad <= 0) { *cos_t_out = 0; refr[0] = refr[1] = refr[2] = 0; return 1; }
    R cos_t = Mx<R>::sqrt_pos(rad);
    R a = n1 * cos_i, b = n2 * cos_t, c = n1 * cos_t, e = n2 * cos_i;
    const R sab = a + b, sce = c + e;
    const R rr = Mx<R>::rcp48(sab * sce);               // both amplitude quotients from one reciprocal (see boundary_planar)
    R rs = (a - b) * sce * rr, rp = (c - e) * sab * rr;
    R k = Nr * cos_i - cos_t;
    refr[0] = d[0] * Nr + nf[0] * k;
    refr[1] = d[1] * Nr + nf[1] * k;
    refr[2] = d[2] * Nr + nf[2] * k;
    const R inv = Mx<R>::rsqrt_pos(refr[0] * refr[0] + refr[1] * refr[1] + refr[2] * refr[2]);   // normalize, S/vectors.py:6-7
    refr[0] *= inv; refr[1] *= inv; refr[2] *= inv;
    *cos_t_out = cos_t;
    return (R)0.5 * (rs * rs + rp * rp);
}

// The same event at a PLANE z = const (layered slabs), where it specialises exactly: cos_i = |uz|; the reflected direction is
// (ux, uy, -uz) -- v - 2 (v.n) n with n = (0, 0, +-1), no rounding at all; the refracted one is (Nr ux, Nr uy, sign(uz) cos_t),
// of unit length by Snell's law (Nr^2 (1 - uz^2) + cos_t^2 = 1), so neither is renormalised.  n1 / n2 and Nr = n1 / n2 come
// from the interface table (IfD).  Returns the unpolarised Fresnel reflectance; TIR (radicand <= 0, :110) returns 1.
// (The CPU checker restates the same operations with IEEE division and square root.)
template <typename R> LT_DEV R boundary_planar(R uz, R n1, R n2, R Nr, R* cos_t_out)
{
    const R cos_i = Mx<R>::abs(uz);
    if (n1 == n2) { *cos_t_out = cos_i; return 0; }
    const R rad = (R)1 - Nr * Nr * ((R)1 - cos_i * cos_i);
    if (rad <= 0) { *cos_t_out = 0; return 1; }
    const R cos_t = Mx<R>::sqrt_pos(rad);
    const R a = n1 * cos_i, b = n2 * cos_t, c = n1 * cos_t, e = n2 * cos_i;
    // both amplitude quotients from ONE reciprocal, refined once (~48 bits): the reflectance is only ever compared with a
    // uniform draw, so 3e-15 of it decides nothing in 1e14 boundary events
    const R sab = a + b, sce = c + e;
    const R r = Mx<R>::rcp48(sab * sce);
    const R rs = (a - b) * sce * r, rp = (c - e) * sab * r;
    *cos_t_out = cos_t;
    return (R)0.5 * (rs * rs + rp * rp);
}

// Spin (App. C.6): MCML direction update, |uz| > 0.99999 special case.
template <typename R, bool TABLE> LT_DEV void spin(R* u, R ct, typename HeldU<R, TABLE>::type xi_phi, const double* T)      // xi_phi: the azimuth's uniform as the walk carries it (f64 XORWOW: the raw draw); T: kWalkMathTab or its LDS copy
{
    const R st2 = Mx<R>::max0((R)1 - ct * ct);          // sin^2(theta); a rounding-negative value is 0
    R sp, cp; HeldU<R, TABLE>::sincos_turn(xi_phi, T, &sp, &cp);
    R ux = u[0], uy = u[1], uz = u[2];
    if (Mx<R>::abs(uz) > (R)0.99999) {
        const R st = Mx<R>::sqrt_unit(st2);
        u[0] = st * cp; u[1] = st * sp; u[2] = uz >= 0 ? ct : -ct;
    } else {
        // MCML's update with A = sin(theta) / sqrt(1 - uz^2) factored out (10 products / fused products instead of 17):
        //   ux' = ux (uz A cos(phi) + cos(theta)) - uy A sin(phi),   uy' = uy (...) + ux A sin(phi),
        //   uz' = uz cos(theta) - A cos(phi) (1 - uz^2)
        // and A from ONE reciprocal root: A = st2 / sqrt(st2 t2) (st2 = 0 gives 0: the seed of 0 is taken at 1e-300)
        const R t2 = (R)1 - uz * uz;                    // |uz| <= 0.99999: t2 >= 2e-5
        const R a = st2 * Mx<R>::rsqrt_pos(Mx<R>::max_tiny(st2 * t2));
        const R sc = a * cp, ss = a * sp;
        const R k = uz * sc + ct;
        u[0] = ux * k - uy * ss;
        u[1] = uy * k + ux * ss;
        u[2] = uz * ct - sc * t2;
    }
}

// ---------------------------------------------------------------------------
// RNG: rocRAND XORWOW, one stream per photon
// ---------------------------------------------------------------------------
LT_DEV unsigned long long mix64(unsigned long long z)
{   // splitmix64's output function
    z = (z ^ (z >> 30)) * 0xBF58476D1CE4E5B9ull;
    z = (z ^ (z >> 27)) * 0x94D049BB133111EBull;
    return z ^ (z >> 31);
}
// One stream per (seed, photon id), hashed in two stages -- the seed first, then the id into the hashed seed -- so that
// no pair (seed + k c, id - k) shares a stream with (seed, id), as a single hash of seed + (id + 1) c would.
LT_DEV unsigned long long mix_seed(unsigned long long seed, unsigned long long photon_id)
{
    return mix64(mix64(seed + 0x9E3779B97F4A7C15ull) ^ ((photon_id + 1ull) * 0x9E3779B97F4A7C15ull));
}

// One stream per photon.  rocrand_init from a seed alone leaves three of the five XORWOW state words offset by the
// same 32-bit value, and every photon's first draws (first step length, first deflection) inherit that structure:
// at 10^8 photons the matched-slab benchmark sat 9e-5 (10 sigma) off van de Hulst's Rd / Tt, with a seed-to-seed
// scatter well below the statistical one.  Discarding kRngWarmup outputs after seeding removes both (4 ... 32
// discards agree to 1e-5; profiles/r01e_rng_seeding.log).  56 instructions per photon against ~10^5 for its walk.
constexpr int kRngWarmup = 8;
LT_DEV void photon_stream(unsigned long long seed, unsigned long long photon_id, rocrand_state_xorwow* st)
{
    rocrand_init(mix_seed(seed, photon_id), 0ull, 0ull, st);
#pragma unroll
    for (int k = 0; k < kRngWarmup; k++) (void)rocrand(st);
}

// ---------------------------------------------------------------------------
// tally
// ---------------------------------------------------------------------------
// value type a deposit is accumulated in before it is sent to memory
template <int TALLY> struct TallyT { typedef double type; };
template <> struct TallyT<LT_TALLY_F32> { typedef float type; };
template <> struct TallyT<LT_TALLY_U64FX> { typedef unsigned long long type; };
#define LT_TALLY_NONE 3  /* diagnostic build of the walk without deposition (not part of the ABI) */

template <int TALLY, typename R> LT_DEV typename TallyT<TALLY>::type tally_quantum(R dw)
{
    if constexpr (TALLY == LT_TALLY_U64FX) return (unsigned long long)((double)dw * LT_FX_SCALE + 0.5);
    else return (typename TallyT<TALLY>::type)dw;
}

template <int TALLY> LT_DEV void tally_add(void* grid, unsigned idx, typename TallyT<TALLY>::type v)
{
    if constexpr (TALLY == LT_TALLY_NONE) { asm volatile("" ::"v"(idx), "v"(v)); }
    else
        __hip_atomic_fetch_add(reinterpret_cast<typename TallyT<TALLY>::type*>(grid) + idx, v, __ATOMIC_RELAXED,
                               __HIP_MEMORY_SCOPE_AGENT);
}

// ---------------------------------------------------------------------------
// LDS image of the scene tables
// ---------------------------------------------------------------------------
LT_DEV size_t align16(size_t x) { return (x + 15) & ~(size_t)15; }

template <typename R> struct LdsLayout {
    size_t off_cnt, off_frame, off_media, off_zb, off_if, off_lm, off_tris, off_nodes, off_links, off_hist, off_march, off_math, total;
    __host__ __device__ LdsLayout(int n_media, int n_layers, int n_tris, int n_nodes, unsigned n_hist = 0, size_t march_bytes = 0)
    {
        auto al = [](size_t x) { return (x + 15) & ~(size_t)15; };
        size_t o = 0;
        off_cnt = o;   o = al(o + 8 * sizeof(double));
        off_frame = o; o = al(o + 15 * sizeof(R));     // frame of the source normal (area sources) + the grid's origin / inverse voxel / dimensions
        off_media = o; o = al(o + (size_t)n_media * sizeof(MedD<R>));
        off_zb = o;    o = al(o + (size_t)(n_layers + 1) * sizeof(R));
        off_if = o;    o = al(o + (size_t)(n_layers > 0 ? n_layers + 1 : 0) * sizeof(IfD<R>));
        off_lm = o;    o = al(o + (size_t)(n_layers > 0 ? n_layers : 1) * sizeof(int32_t));
        off_tris = o;  o = al(o + (size_t)n_tris * sizeof(TriD<R>));
        off_nodes = o; o = al(o + (size_t)n_nodes * sizeof(NodeD<R>));
        off_links = o; o = al(o + (size_t)n_nodes * 16 * sizeof(int16_t));      // front-to-back link tables (8 patterns x first | after)
        off_hist = o;  o = al(o + (size_t)n_hist * sizeof(uint32_t));
        off_march = o; o = al(o + march_bytes);
        off_math = o;  o = al(o + (sizeof(R) == 8 ? kWalkMathTabBytes : 0));      // f64 walks: kWalkMathTab
        total = o;
    }
};

LT_DEV void lds_copy(void* dst, const void* src, size_t bytes)
{
    // tables are small (KBs); dword copies, all threads of the workgroup
    uint32_t* d = reinterpret_cast<uint32_t*>(dst);
    const uint32_t* s = reinterpret_cast<const uint32_t*>(src);
    for (size_t i = threadIdx.x; i < bytes / 4; i += blockDim.x) d[i] = s[i];
}

LT_DEV unsigned long long readlane64(unsigned long long v, int lane)
{
    unsigned lo = __builtin_amdgcn_readlane((unsigned)v, lane);
    unsigned hi = __builtin_amdgcn_readlane((unsigned)(v >> 32), lane);
    return ((unsigned long long)hi << 32) | lo;
}

template <typename T> LT_DEV T wave_sum(T v)
{
#pragma unroll
    for (int off = 32; off > 0; off >>= 1) v += __shfl_down(v, off, 64);
    return v;
}

// light sub-path capture (f4): store vertex k of photon `rel` while k < max_vertices
template <typename R>
LT_DEV void record_vertex(const WalkParams& P, unsigned long long rel, unsigned& nv, R px, R py, R pz, R ux, R uy,
                          R uz, R w, int kind, int medium, unsigned step, R gx, R gy, R gz, R pdf_pos, R pdf_dir)
{
    if (nv < P.max_vertices) {
        lt_vertex* v = P.vertices + rel * (unsigned long long)P.max_vertices + nv;
        v->point[0] = (double)px; v->point[1] = (double)py; v->point[2] = (double)pz;
        v->direction[0] = (double)ux; v->direction[1] = (double)uy; v->direction[2] = (double)uz;
        v->g_norm[0] = (double)gx; v->g_norm[1] = (double)gy; v->g_norm[2] = (double)gz;
        v->throughput = (double)w; v->pdf_pos = (double)pdf_pos; v->pdf_dir = (double)pdf_dir;
        v->kind = kind; v->medium = medium; v->step = step; v->pad_ = 0;
        nv++;
        P.vertex_counts[rel] = nv;
    }
}

// One deposit record per lane (or none) leaves the lane's run-length accumulator.  Atomic mode: a no-return
// global atomic.  Log mode: the wave's records are compacted by ballot rank and appended, coalesced, to the wave's
// current log chunk (SoA: voxel index, value); a chunk is claimed with one returning atomic per kLogChunk records.
// If the log is exhausted the wave falls back to atomics, so a too-small log costs speed, never correctness.
constexpr unsigned kLogExhausted = 0xfffffffeu;   // lg_chunk: this wave has found the log full (0xffffffff: no chunk claimed yet)
template <int TALLY>
LT_DEV void emit_deposit(const WalkParams& P, bool has, unsigned idx, typename TallyT<TALLY>::type val,
                         unsigned& lg_cur, unsigned& lg_end, unsigned& lg_chunk, uint32_t* s_hist)
{
    typedef typename TallyT<TALLY>::type TV;
    if (P.log_idx == nullptr) {
        if (has) tally_add<TALLY>(P.grid, idx, val);
        return;
    }
    const unsigned long long m = __ballot(has);
    if (m == 0ull) return;
    const unsigned cnt = (unsigned)__popcll(m);
    // (a wave that has found the log exhausted does not ask again: lg_chunk == kLogExhausted is sticky, so an undersized log
    // costs each wave ONE extra returning atomic, not one per emit on 16 shared addresses, and the group counters stay
    // within cap + resident waves -- far from wrapping grp + kLogGroups * c)
    if (lg_chunk != kLogExhausted && lg_end - lg_cur < cnt) {
        const int lane = threadIdx.x & 63;
        const unsigned grp = blockIdx.x & (kLogGroups - 1);
        if (lg_chunk < kLogExhausted && lane == 0) P.log_fill[lg_chunk] = lg_cur - lg_chunk * kLogChunk;
        unsigned c = 0;
        if (lane == 0) c = __hip_atomic_fetch_add(P.log_next + grp, 1u, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        c = grp + kLogGroups * __builtin_amdgcn_readfirstlane(c);     // the group's own chunks: see kLogGroups
        if (c < P.log_cap_chunks) { lg_chunk = c; lg_cur = c * kLogChunk; lg_end = lg_cur + kLogChunk; }
        else { lg_chunk = kLogExhausted; lg_cur = lg_end = 0; }
    }
    if (lg_chunk == kLogExhausted) {  // log exhausted: back to the linear voxel index and a global atomic
        if ((threadIdx.x & 63) == 0) __hip_atomic_fetch_add(P.log_overflow, cnt, __ATOMIC_RELAXED, __HIP_MEMORY_SCOPE_AGENT);
        if (has) {
            // (the divisors pass through an empty asm so that the reciprocal set-up of these divisions stays in this rare
            // branch: left visible as loop invariants it was hoisted in front of the walk's loop and parked in registers)
            unsigned ntx = P.log_ntx, nty = P.log_nty;
            asm volatile("" : "+s"(ntx), "+s"(nty));
            const unsigned tile = idx >> kTileShift, tx = tile % ntx, ty = (tile / ntx) % nty,
                           tz = tile / (ntx * nty);
            const unsigned vx = (tx << kTileBX) | (idx & 31u), vy = (ty << kTileBY) | ((idx >> 5) & 31u),
                           vz = (tz << kTileBZ) | ((idx >> 10) & 15u);
            tally_add<TALLY>(P.grid, (vz * (unsigned)P.ny + vy) * (unsigned)P.nx + vx, val);
        }
        return;
    }
    if (has) {
        const unsigned rank = __builtin_amdgcn_mbcnt_hi((unsigned)(m >> 32), __builtin_amdgcn_mbcnt_lo((unsigned)m, 0u));
        P.log_idx[lg_cur + rank] = idx;
        reinterpret_cast<TV*>(P.log_val)[lg_cur + rank] = val;
        atomicAdd(&s_hist[idx >> P.log_hist_shift], 1u);   // histogram of the first partition pass's bins, kept in LDS
    }
    lg_cur += cnt;
}

constexpr unsigned kPacket = 64;  // photon ids taken from the global queue per atomic
#ifndef LT_REFILL_MIN
#define LT_REFILL_MIN 4
#endif
constexpr unsigned kRefillMin = LT_REFILL_MIN;   // dead lanes a wave collects before it refills them
#ifndef LT_QUERY_MIN
#define LT_QUERY_MIN 8
#endif
// mesh walks: lanes a wave collects before it serves their surface queries.  16 was best while every query walked the
// BVH (~1500 instructions per service); with the near-triangle lists (nearest_listed: ~2-4 triangle tests) a service is
// cheap and waiting costs more: C4 f64 walk 37.3 ms at 16, 34.4 at 8, 34.5 at 4, 47.6 at 32 (profiles/r02d_c4_near_lists.log)
constexpr unsigned kQueryMin = LT_QUERY_MIN;
// ... and when the tables live in global memory (GEOM 2): there a service that has to walk the BVH for some lane is a
// chain of dependent loads that costs the wave the same whether 2 lanes or 20 walk, so it pays to wait for many: on the
// 5140-triangle sphere 8 / 16 / 24 / 32 / 48 / 56 / 64 lanes give 4.7 / 5.4 / 6.3 / 7.1 / 8.5 / 8.2 / 5.2e9
// photon-steps/s f64 (profiles/r02e_large_mesh_query_threshold.log).  LT_QUERY_MIN in the environment overrides both.
constexpr unsigned kQueryMinGlobal = 48;
#ifndef LT_QUERY_MAX_AGE
#define LT_QUERY_MAX_AGE 8
#endif
// ... but no lane waits longer than this many wave iterations: where queries are rare (a dense medium around a large mesh)
// the lanes that wait would idle until enough of them have trickled in -- the 5140-triangle sphere in a medium of ten
// times the scattering ran at 4.2e9 photon-steps/s without this bound and at 9.1-9.5e9 with 4 ... 10; C4 gains 1-5 % too
// (profiles/r02e_large_mesh_query_threshold.log)
constexpr unsigned kQueryMaxAge = LT_QUERY_MAX_AGE;
#ifndef LT_MARCH_DRAIN_MIN
#define LT_MARCH_DRAIN_MIN 8
#endif
// walk_kernel_m: queries whose march has ended and that only wait for their queued candidates to be tested; this many
// trigger a drain of a queue that holds less than a full round (LT_QUERY_MIN in the environment overrides)
constexpr unsigned kMarchDrainMin = LT_MARCH_DRAIN_MIN;

// ---------------------------------------------------------------------------
// the walk kernel
// ---------------------------------------------------------------------------
// minimum waves per SIMD the register allocator must leave room for (occupancy is what hides the latency of the
// dependent f64 / transcendental chains once deposition no longer paces the walk)
#ifndef LT_F64_WAVES
#define LT_F64_WAVES 4        // f64 mesh walks: ~80 B of scratch per lane at 128 VGPRs, still 6 % faster than 3 waves (C4)
#endif
#ifndef LT_F64_GLOBAL_WAVES
#define LT_F64_GLOBAL_WAVES 3 // f64 walks of meshes beyond LDS (grid march): the DDA state on top of the photon's does not fit 128 VGPRs
#endif
#ifndef LT_F64_SLAB_WAVES
#define LT_F64_SLAB_WAVES 4   // f64 slab walks fit 128 VGPRs without scratch since the polynomial constants live in SGPRs
#endif
#ifndef LT_F32_WAVES
#define LT_F32_WAVES 5
#endif
// where the walk keeps the grid's origin / inverse voxel size / dimensions (lt_walk_kernel.inc, LT_INV_MODE), chosen per kernel
// from same-box A/B runs (profiles/r04c_ab_modes.log):
//   slab kernels   1, pinned VGPRs: f64 127 VGPRs, the fewest VALU instructions (C2 walk 26.3 ms against 26.9 converted in
//                  the kernel, 27.0 as scalar arguments, 27.2 from LDS) -- and the leaner builds (109-110 VGPRs) made the log
//                  reduction beside a half-occupancy walk much slower (two lanes: 40.6-42.5 ms against 37.5-37.9)
//   LDS-mesh       2, LDS: the Cornell kernel needs its 128 VGPRs for the photon and the held step; no scratch (60 B pinned),
//                  C4 walk 28.4 ms against 31.7 pinned and 33.5 in round 3
//   march kernel   2, LDS: teapot / cow / pumpkin 19.0 / 18.9 / 17.8e9 f64 steps/s against 17.8 / 17.6 / 16.9 as scalar arguments
//                  (156 VGPRs without scratch, but the scalar file overflows into v_readlane restores) and 18.1 / 18.0 / 17.1
//                  pinned (profiles/r04d_march_modes.log); 154 VGPRs + 36 B of scratch
#ifndef LT_INV_MODE_SLAB
#define LT_INV_MODE_SLAB 1
#endif
#ifndef LT_INV_MODE_MESH
#define LT_INV_MODE_MESH 2
#endif
#ifndef LT_INV_MODE_MARCH
#define LT_INV_MODE_MARCH 2
#endif
#define LT_INV_MODE LT_INV_MODE_SLAB
#define LT_WALK_NAME walk_kernel
#define LT_WALK_BATCH 0
#include "lt_walk_kernel.inc"
#undef LT_WALK_NAME
#undef LT_WALK_BATCH
#undef LT_INV_MODE
#define LT_INV_MODE LT_INV_MODE_MESH
#define LT_WALK_NAME walk_kernel_q
#define LT_WALK_BATCH 1
#include "lt_walk_kernel.inc"
#undef LT_WALK_NAME
#undef LT_WALK_BATCH
#undef LT_INV_MODE
#define LT_INV_MODE LT_INV_MODE_MARCH
#define LT_WALK_NAME walk_kernel_m
#define LT_WALK_BATCH 2
#include "lt_walk_kernel.inc"
#undef LT_WALK_NAME
#undef LT_WALK_BATCH
#undef LT_INV_MODE

// ---------------------------------------------------------------------------
// variant dispatch
// ---------------------------------------------------------------------------
typedef void (*WalkFn)(const WalkParams);

// CAPTURE (light sub-path vertices, f4) is a compile-time variant: the capture code costs the plain walk registers
// even behind a never-taken branch (36-84 B of scratch per lane, walk 29.7 -> 34.6 ms when it was a run-time test).
// Captures run the f64 walk with the XORWOW generator.
template <typename R, int GEOM, bool TABLE, bool CAPTURE>
static WalkFn pick_tally(int tally)
{
    switch (tally) {
    // slabs: walk_kernel; meshes: walk_kernel_q (the same body with batched BVH queries, lt_walk_kernel.inc)
    case LT_TALLY_F32:
        if constexpr (TABLE) return nullptr;
        else if constexpr (GEOM == 0) return walk_kernel<R, GEOM, TABLE, LT_TALLY_F32, CAPTURE>;
        else return walk_kernel_q<R, GEOM, TABLE, LT_TALLY_F32, CAPTURE>;
    case LT_TALLY_F64:
        if constexpr (GEOM == 0) return walk_kernel<R, GEOM, TABLE, LT_TALLY_F64, CAPTURE>;
        else return walk_kernel_q<R, GEOM, TABLE, LT_TALLY_F64, CAPTURE>;
    case LT_TALLY_U64FX:
        if constexpr (GEOM == 0) return walk_kernel<R, GEOM, TABLE, LT_TALLY_U64FX, CAPTURE>;
        else return walk_kernel_q<R, GEOM, TABLE, LT_TALLY_U64FX, CAPTURE>;
    case LT_TALLY_NONE:
        if constexpr (!TABLE && GEOM == 0 && !CAPTURE) return walk_kernel<R, GEOM, TABLE, LT_TALLY_NONE, false>; else return nullptr;
    }
    return nullptr;
}

// meshes beyond LDS with a march grid (Variant::mesh 3): walk_kernel_m, tables in global memory (GEOM 2)
template <typename R, bool CAPTURE>
static WalkFn pick_march(int tally)
{
    switch (tally) {
    case LT_TALLY_F32: return walk_kernel_m<R, 2, false, LT_TALLY_F32, CAPTURE>;
    case LT_TALLY_F64: return walk_kernel_m<R, 2, false, LT_TALLY_F64, CAPTURE>;
    case LT_TALLY_U64FX: return walk_kernel_m<R, 2, false, LT_TALLY_U64FX, CAPTURE>;
    }
    return nullptr;
}

template <typename R, bool TABLE, bool CAPTURE>
static WalkFn pick_geom(const Variant& v)
{
    if (v.mesh == 3) { if constexpr (!TABLE) return pick_march<R, CAPTURE>(v.tally); else return nullptr; }
    if (v.mesh == 0) return pick_tally<R, 0, TABLE, CAPTURE>(v.tally);
    if (v.mesh == 1) return pick_tally<R, 1, TABLE, CAPTURE>(v.tally);
    if constexpr (!TABLE) return pick_tally<R, 2, TABLE, CAPTURE>(v.tally); else return nullptr;
}

// slab walks in two kernels (Variant::phase 1 / 2, lt_walk_kernel.inc "Tail split"): XORWOW, no capture
template <typename R, int PHASE>
static WalkFn pick_phase(int tally)
{
    switch (tally) {
    case LT_TALLY_F32: return walk_kernel<R, 0, false, LT_TALLY_F32, false, PHASE>;
    case LT_TALLY_F64: return walk_kernel<R, 0, false, LT_TALLY_F64, false, PHASE>;
    case LT_TALLY_U64FX: return walk_kernel<R, 0, false, LT_TALLY_U64FX, false, PHASE>;
    }
    return nullptr;
}

// ... and the walk through a mesh in LDS (Variant::mesh 1)
template <typename R, int PHASE>
static WalkFn pick_phase_q(int tally)
{
    switch (tally) {
    case LT_TALLY_F32: return walk_kernel_q<R, 1, false, LT_TALLY_F32, false, PHASE>;
    case LT_TALLY_F64: return walk_kernel_q<R, 1, false, LT_TALLY_F64, false, PHASE>;
    case LT_TALLY_U64FX: return walk_kernel_q<R, 1, false, LT_TALLY_U64FX, false, PHASE>;
    }
    return nullptr;
}

static WalkFn pick(const Variant& v)
{
    if (v.phase) {
        if (v.mesh > 1 || v.table || v.capture) return nullptr;
        if (v.mesh == 1) {
            if (v.phase == 1) return v.f32 ? pick_phase_q<float, 1>(v.tally) : pick_phase_q<double, 1>(v.tally);
            return v.f32 ? pick_phase_q<float, 2>(v.tally) : pick_phase_q<double, 2>(v.tally);
        }
        if (v.phase == 1) return v.f32 ? pick_phase<float, 1>(v.tally) : pick_phase<double, 1>(v.tally);
        return v.f32 ? pick_phase<float, 2>(v.tally) : pick_phase<double, 2>(v.tally);
    }
    if (v.capture) return (v.f32 || v.table) ? nullptr : pick_geom<double, false, true>(v);
    if (v.f32) return v.table ? nullptr : pick_geom<float, false, false>(v);
    return v.table ? pick_geom<double, true, false>(v) : pick_geom<double, false, false>(v);
}

size_t walk_lds_bytes(const Variant& v, int n_media, int n_layers, int n_tris, int n_nodes, unsigned n_hist)
{
    if (v.mesh) n_layers = 0;
    if (v.mesh != 1) { n_tris = 0; n_nodes = 0; }
    // walk_kernel_m (mesh 3): the march's per-wave scratch (rays, best hits, candidate queue), four waves per workgroup.  The
    // plain global-memory kernel (mesh 2: no march grid / LT_NO_MARCH) never touches it and keeps its LDS free
    return v.f32 ? LdsLayout<float>(n_media, n_layers, n_tris, n_nodes, n_hist, v.mesh == 3 ? 4 * sizeof(MarchWave<float>) : 0).total
                 : LdsLayout<double>(n_media, n_layers, n_tris, n_nodes, n_hist, v.mesh == 3 ? 4 * sizeof(MarchWave<double>) : 0).total;
}

int walk_max_blocks_per_cu(const Variant& v, int threads, size_t lds_bytes)
{
    WalkFn fn = pick(v);
    if (!fn) return 0;
    int nb = 0;
    if (hipOccupancyMaxActiveBlocksPerMultiprocessor(&nb, reinterpret_cast<const void*>(fn), threads, lds_bytes) != hipSuccess)
        return 0;
    return nb;
}

hipError_t launch_walk(const WalkParams& Pin, const Variant& v, const LaunchCfg& cfg, hipStream_t s)
{
    WalkFn fn = pick(v);
    if (!fn) return hipErrorInvalidValue;
    WalkParams P = Pin;
    if (v.mesh) P.n_layers = 0; else { P.n_tris = 0; P.n_nodes = 0; }
    if (cfg.lds_bytes > 64 * 1024) {
        hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(fn),
                                           hipFuncAttributeMaxDynamicSharedMemorySize, (int)cfg.lds_bytes);
        if (e != hipSuccess) return e;
    }
    hipLaunchKernelGGL(fn, dim3(cfg.blocks), dim3(cfg.threads), cfg.lds_bytes, s, P);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// query kernels: the same __device__ functions, one lane per query (f64)
// ---------------------------------------------------------------------------
__global__ void k_intersect_rays(const TriD<double>* tris, const NodeD<double>* nodes, int n_tris, int n_nodes,
                                 const double* o, const double* d, const double* tmax, size_t n, int use_bvh,
                                 const MarchGrid G, const int16_t* links, int32_t* prim, double* t)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (use_bvh == 2) {     // the walk's wave-cooperative march (256 threads per workgroup: four waves)
        __shared__ MarchWave<double> mw[4];
        const bool want = i < n;
        const size_t k = want ? i : 0;
        int pi; double tt;
        march_service<double>(tris, nodes, n_nodes, G, &mw[threadIdx.x >> 6], want, o[3 * k], o[3 * k + 1], o[3 * k + 2], d[3 * k], d[3 * k + 1],
                              d[3 * k + 2], tmax ? tmax[k] : __builtin_huge_val(), 0.0, pi, tt);
        if (want) { prim[i] = pi; t[i] = tt; }
        return;
    }
    if (i >= n) return;
    double oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, dd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    double tm = tmax ? tmax[i] : __builtin_huge_val();
    int pi; double tt;
    if (use_bvh == 4) nearest_bvh_ordered(tris, nodes, n_nodes, links + ray_octant(dd) * 2 * n_nodes, oo, dd, tm, pi, tt);   // front to back (the renderers' order)
    else if (use_bvh == 3) nearest_march(tris, nodes, n_nodes, G, oo, dd, tm, 0.0, pi, tt);      // the same march, lane by lane
    else if (use_bvh) nearest_bvh(tris, nodes, n_nodes, oo, dd, tm, pi, tt);
    else nearest_brute(tris, n_tris, oo, dd, tm, pi, tt);
    prim[i] = pi; t[i] = tt;
}

__global__ void k_triangle_intersect(const double* o, const double* d, const double* tris, size_t n, double* t)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    // derive the PreComputedTriangle fields exactly as the host does
    // (S/primitives.py:105-111), without fused multiply-adds
    const double* a = tris + 9 * i; const double* b = a + 3; const double* c = a + 6;
    TriD<double> T;
    {
#pragma clang fp contract(off)
        for (int k = 0; k < 3; k++) { T.a[k] = a[k]; T.ab[k] = b[k] - a[k]; T.ac[k] = c[k] - a[k]; }
        double nn[3];
        nn[0] = T.ab[1] * T.ac[2] - T.ab[2] * T.ac[1];
        nn[1] = T.ab[2] * T.ac[0] - T.ab[0] * T.ac[2];
        nn[2] = T.ab[0] * T.ac[1] - T.ab[1] * T.ac[0];
        double l = ::sqrt(nn[0] * nn[0] + nn[1] * nn[1] + nn[2] * nn[2]);
        for (int k = 0; k < 3; k++) T.n[k] = nn[k] / l;
    }
    T.med_front = T.med_back = -1;
    double oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]}, dd[3] = {d[3 * i], d[3 * i + 1], d[3 * i + 2]};
    t[i] = tri_hit(oo, dd, &T);
}

__global__ void k_intersect_bounds(const double* o, const double* d, const double* tmax, const double* boxes,
                                   size_t n, int32_t* hit)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    double oo[3] = {o[3 * i], o[3 * i + 1], o[3 * i + 2]};
    double inv[3] = {1.0 / d[3 * i], 1.0 / d[3 * i + 1], 1.0 / d[3 * i + 2]};
    double lo[3] = {boxes[6 * i], boxes[6 * i + 1], boxes[6 * i + 2]};
    double hi[3] = {boxes[6 * i + 3], boxes[6 * i + 4], boxes[6 * i + 5]};
    hit[i] = box_hit(lo, hi, oo, inv, tmax ? tmax[i] : __builtin_huge_val()) ? 1 : 0;
}

__global__ void k_eval(int fn, const double* in, size_t n, double* out)
{
    size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= n) return;
    switch (fn) {
    case LT_FN_HG_PDF: out[i] = hg_pdf(in[2 * i], in[2 * i + 1]); break;
    case LT_FN_HG_SAMPLE: {
        double g = in[2 * i + 1];
        out[i] = hg_sample(in[2 * i], g, 1.0 - g * g, 1.0 + g * g, 1.0 / (2.0 * g));
    } break;
    case LT_FN_ONB: {
        double nn[3] = {in[3 * i], in[3 * i + 1], in[3 * i + 2]}, v2[3], v3[3];
        onb(nn, v2, v3);
        for (int k = 0; k < 3; k++) { out[6 * i + k] = v2[k]; out[6 * i + 3 + k] = v3[k]; }
    } break;
    case LT_FN_DISK: { double dd[2]; disk(in[2 * i], in[2 * i + 1], dd); out[2 * i] = dd[0]; out[2 * i + 1] = dd[1]; } break;
    case LT_FN_COSINE_HEMI: {
        double nn[3] = {in[8 * i], in[8 * i + 1], in[8 * i + 2]}, wi[3] = {in[8 * i + 3], in[8 * i + 4], in[8 * i + 5]}, o4[4];
        cosine_hemi(nn, wi, in[8 * i + 6], in[8 * i + 7], o4);
        for (int k = 0; k < 4; k++) out[4 * i + k] = o4[k];
    } break;
    case LT_FN_REFLECT: {
        double v[3] = {in[6 * i], in[6 * i + 1], in[6 * i + 2]}, nn[3] = {in[6 * i + 3], in[6 * i + 4], in[6 * i + 5]}, r[3];
        reflect(v, nn, r);
        for (int k = 0; k < 3; k++) out[3 * i + k] = r[k];
    } break;
    case LT_FN_BOUNDARY: {
        double dd[3] = {in[8 * i], in[8 * i + 1], in[8 * i + 2]}, nf[3] = {in[8 * i + 3], in[8 * i + 4], in[8 * i + 5]}, ct, refr[3];
        out[5 * i] = boundary(dd, nf, in[8 * i + 6], in[8 * i + 7], &ct, refr);
        out[5 * i + 1] = ct; out[5 * i + 2] = refr[0]; out[5 * i + 3] = refr[1]; out[5 * i + 4] = refr[2];
    } break;
    case LT_FN_SPIN: {
        double u[3] = {in[5 * i], in[5 * i + 1], in[5 * i + 2]};
        spin<double, true>(u, in[5 * i + 3], in[5 * i + 4], kWalkMathTab);
        for (int k = 0; k < 3; k++) out[3 * i + k] = u[k];
    } break;
    case LT_FN_WALK_MATH: {   // the walk's lean f64 primitives: -ln x, sin/cos(2 pi x), sqrt x, 1 / (1 + x)
        const double x = in[i];
        out[5 * i] = neg_log_tab(x, kWalkMathTab);
        sincos_turn_tab(x, kWalkMathTab, &out[5 * i + 1], &out[5 * i + 2]);
        out[5 * i + 3] = sqrt01(x);
        out[5 * i + 4] = fast_div(1.0, 1.0 + x);
    } break;
    case LT_FN_WALK_MATH_RAW: {   // ... and the forms the f64 XORWOW walk applies to the raw draw
        const unsigned k = (unsigned)in[i];
        out[3 * i] = neg_log_raw(k, kWalkMathTab);
        sincos_turn_raw(k, kWalkMathTab, &out[3 * i + 1], &out[3 * i + 2]);
    } break;
    }
}

__global__ void k_rng_raw(unsigned long long seed, unsigned long long photon_id, unsigned count, uint32_t* out)
{
    if (blockIdx.x != 0 || threadIdx.x != 0) return;
    rocrand_state_xorwow st;
    photon_stream(seed, photon_id, &st);
    for (unsigned i = 0; i < count; i++) out[i] = rocrand(&st);
}

__global__ void k_grid_to_f64(const void* grid, int tally, size_t n, double* out)
{
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) {
        if (tally == LT_TALLY_F32) out[i] = (double)reinterpret_cast<const float*>(grid)[i];
        else if (tally == LT_TALLY_F64) out[i] = reinterpret_cast<const double*>(grid)[i];
        else out[i] = (double)reinterpret_cast<const unsigned long long*>(grid)[i] * (1.0 / LT_FX_SCALE);
    }
}

// dst += src over whole grids, 16 bytes per lane (the grids are hipMalloc'ed: 256-byte aligned).  Joins the private
// grid of an overlapped launch's second lane into the ctx grid: HBM-bound, 3 x grid bytes.
template <typename T>
__global__ void __launch_bounds__(256) k_grid_add(T* __restrict__ dst, const T* __restrict__ src, size_t n)
{
    constexpr size_t kV = 16 / sizeof(T);
    struct alignas(16) V { T v[kV]; };
    const size_t nv = n / kV;
    for (size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < nv; i += (size_t)gridDim.x * blockDim.x) {
        V a = reinterpret_cast<V*>(dst)[i];
        const V b = reinterpret_cast<const V*>(src)[i];
#pragma unroll
        for (size_t k = 0; k < kV; k++) a.v[k] += b.v[k];
        reinterpret_cast<V*>(dst)[i] = a;
    }
    for (size_t i = nv * kV + (size_t)blockIdx.x * blockDim.x + threadIdx.x; i < n; i += (size_t)gridDim.x * blockDim.x) dst[i] += src[i];
}

// ---------------------------------------------------------------------------
// clearance grid builder: per cell, min over triangles of the distance from the cell centre to the triangle
// (closest-point-on-triangle by regions), minus the half diagonal of the cell, shrunk by a safety margin
// ---------------------------------------------------------------------------
LT_DEV double point_tri_dist2(const double* p, const TriD<double>& T)
{
    const double ab[3] = {T.ab[0], T.ab[1], T.ab[2]}, ac[3] = {T.ac[0], T.ac[1], T.ac[2]};
    const double ap[3] = {p[0] - T.a[0], p[1] - T.a[1], p[2] - T.a[2]};
    const double d1 = dot3(ab, ap), d2 = dot3(ac, ap);
    double q[3];
    if (d1 <= 0 && d2 <= 0) { q[0] = 0; q[1] = 0; q[2] = 0; }                       // vertex A
    else {
        const double bp[3] = {ap[0] - ab[0], ap[1] - ab[1], ap[2] - ab[2]};
        const double d3 = dot3(ab, bp), d4 = dot3(ac, bp);
        const double cp[3] = {ap[0] - ac[0], ap[1] - ac[1], ap[2] - ac[2]};
        const double d5 = dot3(ab, cp), d6 = dot3(ac, cp);
        const double vc = d1 * d4 - d3 * d2, vb = d5 * d2 - d1 * d6, va = d3 * d6 - d5 * d4;
        if (d3 >= 0 && d4 <= d3) { q[0] = ab[0]; q[1] = ab[1]; q[2] = ab[2]; }   // vertex B
        else if (vc <= 0 && d1 >= 0 && d3 <= 0) { const double v = d1 / (d1 - d3); for (int k = 0; k < 3; k++) q[k] = v * ab[k]; }
        else if (d6 >= 0 && d5 <= d6) { q[0] = ac[0]; q[1] = ac[1]; q[2] = ac[2]; }  // vertex C
        else if (vb <= 0 && d2 >= 0 && d6 <= 0) { const double w = d2 / (d2 - d6); for (int k = 0; k < 3; k++) q[k] = w * ac[k]; }
        else if (va <= 0 && (d4 - d3) >= 0 && (d5 - d6) >= 0) {
            const double w = (d4 - d3) / ((d4 - d3) + (d5 - d6));
            for (int k = 0; k < 3; k++) q[k] = ab[k] + w * (ac[k] - ab[k]);
        } else {
            const double den = 1.0 / (va + vb + vc), v = vb * den, w = vc * den;
            for (int k = 0; k < 3; k++) q[k] = v * ab[k] + w * ac[k];
        }
    }
    const double e[3] = {ap[0] - q[0], ap[1] - q[1], ap[2] - q[2]};
    return dot3(e, e);
}

// f16 <-> f32 for the clearance records (values >= 0; encode rounds toward zero so that the bound stays conservative)
LT_DEV unsigned half_down(double c)
{
    if (!(c > 0)) return 0u;
    if (!(c < 65504.0)) return c == __builtin_huge_val() ? 0x7c00u : 0x7bffu;
    _Float16 h = (_Float16)(float)c;
    unsigned short b = __builtin_bit_cast(unsigned short, h);
    if ((double)(float)h > c && b > 0) b--;
    return b;
}
LT_DEV float half_up(unsigned bits) { return (float)__builtin_bit_cast(_Float16, (unsigned short)(bits & 0xffffu)); }

__global__ void k_build_clearance(const TriD<double>* tris, int n_tris, int near_lists, uint4* clear, int nx, int ny, int nz,
                                  double ox, double oy, double oz, double hx, double hy, double hz)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)nx * ny * nz) return;
    const int ix = (int)(i % nx), iy = (int)((i / nx) % ny), iz = (int)(i / ((size_t)nx * ny));
    const double p[3] = {ox + (ix + 0.5) * hx, oy + (iy + 0.5) * hy, oz + (iz + 0.5) * hz};
    // the five nearest triangles of the cell centre, nearest first (squared distances; ties keep the lower index first)
    const double inf = __builtin_huge_val();
    double best[5] = {inf, inf, inf, inf, inf};
    int id[5] = {-1, -1, -1, -1, -1};
    for (int t = 0; t < n_tris; t++) {
        double d2 = point_tri_dist2(p, tris[t]);
        int ti = t;
#pragma unroll
        for (int k = 0; k < 5; k++)
            if (d2 < best[k]) { const double sd = best[k]; const int si = id[k]; best[k] = d2; id[k] = ti; d2 = sd; ti = si; }
    }
    const double half_diag = 0.5 * ::sqrt(hx * hx + hy * hy + hz * hz);
    auto bound = [&](double d2) -> double {     // strictly conservative for every point of the cell
        if (d2 == inf) return inf;
        const double c = (::sqrt(d2) - half_diag) * (1.0 - 1e-6) - 1e-7 * (hx + hy + hz);
        return c > 0 ? c : 0.0;
    };
    const double c0 = bound(best[0]);
    float f = (float)c0;
    if ((double)f > c0 && f > 0) f = __uint_as_float(__float_as_uint(f) - 1u);      // round toward zero
    if (c0 == inf) f = 3.0e38f;
    unsigned y = 0, z = 0xffffffffu, w = 0xffffffffu;
    if (near_lists) {      // (ids are 16 bits: the host builds these records only for meshes whose tables fit LDS, <= ~1100 triangles, and refuses otherwise)
        y = half_down(bound(best[2])) | (half_down(bound(best[4])) << 16);
        z = (unsigned)(id[0] & 0xffff) | ((unsigned)(id[1] & 0xffff) << 16);      // (-1 & 0xffff = 0xffff: none)
        w = (unsigned)(id[2] & 0xffff) | ((unsigned)(id[3] & 0xffff) << 16);
    }
    clear[i] = make_uint4(__float_as_uint(f), y, z, w);
}

hipError_t launch_build_clearance(const void* tris_f64, int n_tris, int near_lists, void* clear_records, int nx, int ny, int nz,
                                  const double org[3], const double cell[3], hipStream_t s)
{
    const size_t n = (size_t)nx * ny * nz;
    if (n == 0 || n_tris <= 0) return hipSuccess;
    hipLaunchKernelGGL(k_build_clearance, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s,
                       reinterpret_cast<const TriD<double>*>(tris_f64), n_tris, near_lists, reinterpret_cast<uint4*>(clear_records), nx, ny, nz,
                       org[0], org[1], org[2], cell[0], cell[1], cell[2]);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// march grid: c0 of every cell from the EXACT distance of the cell centre to the mesh, found by a nearest-point query in
// the BVH (a subtree is skipped when its box is not nearer than the best triangle so far).  Two levels: a coarse grid
// (cells 8 x 8 x 8 fine cells wide) is searched without a bound; a fine cell then starts from the bound its coarse cell
// implies, dist(p) <= dist(q) + |p - q|, so its search only opens the few subtrees around that sphere.  Cost ~ cells x
// tens of nodes instead of cells x triangles (k_build_clearance): 10^4 triangles get 128^3 cells instead of 58^3.
// ---------------------------------------------------------------------------
LT_DEV double point_box_dist2(const double* p, const NodeD<double>& nd)
{
    double s = 0;
#pragma unroll
    for (int k = 0; k < 3; k++) {
        const double e = __builtin_fmax(__builtin_fmax(nd.lo[k] - p[k], p[k] - nd.hi[k]), 0.0);
        s += e * e;
    }
    return s;
}
LT_DEV double mesh_dist2(const TriD<double>* tris, const NodeD<double>* nodes, int n_nodes, const double* p, double best)
{
    int cur = 0;
    while (cur < n_nodes) {
        const NodeD<double>& nd = nodes[cur];
        if (point_box_dist2(p, nd) <= best) {
            if (nd.n_prims > 0) {
                for (int k = 0; k < nd.n_prims; k++) best = __builtin_fmin(best, point_tri_dist2(p, tris[nd.offset + k]));
                cur = nd.skip;
            } else cur++;
        } else cur = nd.skip;
    }
    return best;
}
__global__ void k_march_coarse(const TriD<double>* tris, const NodeD<double>* nodes, int n_nodes, const MarchGrid G, int kx, int ky,
                               int kz, double* coarse)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)kx * ky * kz) return;
    const int ix = (int)(i % kx), iy = (int)((i / kx) % ky), iz = (int)(i / ((size_t)kx * ky));
    const double p[3] = {G.org[0] + (ix + 0.5) * 8.0 * G.h[0], G.org[1] + (iy + 0.5) * 8.0 * G.h[1], G.org[2] + (iz + 0.5) * 8.0 * G.h[2]};
    coarse[i] = ::sqrt(mesh_dist2(tris, nodes, n_nodes, p, __builtin_huge_val()));
}
__global__ void k_march_fine(const TriD<double>* tris, const NodeD<double>* nodes, int n_nodes, const MarchGrid G, int kx, int ky,
                             const double* coarse, uint2* cells)
{
    const size_t i = (size_t)blockIdx.x * blockDim.x + threadIdx.x;
    if (i >= (size_t)G.nx * G.ny * G.nz) return;
    const int ix = (int)(i % G.nx), iy = (int)((i / G.nx) % G.ny), iz = (int)(i / ((size_t)G.nx * G.ny));
    const double p[3] = {G.org[0] + (ix + 0.5) * G.h[0], G.org[1] + (iy + 0.5) * G.h[1], G.org[2] + (iz + 0.5) * G.h[2]};
    const int qx = ix >> 3, qy = iy >> 3, qz = iz >> 3;
    const double q[3] = {G.org[0] + (qx + 0.5) * 8.0 * G.h[0], G.org[1] + (qy + 0.5) * 8.0 * G.h[1], G.org[2] + (qz + 0.5) * 8.0 * G.h[2]};
    const double e[3] = {p[0] - q[0], p[1] - q[1], p[2] - q[2]};
    const double up = (coarse[((size_t)qz * ky + qy) * kx + qx] + ::sqrt(dot3(e, e))) * (1.0 + 1e-9);     // dist(p) <= this
    const double d2 = mesh_dist2(tris, nodes, n_nodes, p, up * up);
    const double half_diag = 0.5 * ::sqrt(G.h[0] * G.h[0] + G.h[1] * G.h[1] + G.h[2] * G.h[2]);
    // strictly conservative for every point of the cell (as k_build_clearance)
    double c0 = (::sqrt(d2) - half_diag) * (1.0 - 1e-6) - 1e-7 * (G.h[0] + G.h[1] + G.h[2]);
    if (!(c0 > 0)) c0 = 0.0;
    float f = (float)c0;
    if ((double)f > c0 && f > 0) f = __uint_as_float(__float_as_uint(f) - 1u);      // round toward zero
    cells[i].x = (__float_as_uint(f) & ~kMarchCountMask) | (cells[i].x & kMarchCountMask);      // the low bits hold the cell's candidate count
}
hipError_t launch_march_clearance(const void* tris_f64, const void* nodes_f64, int n_nodes, const MarchGrid& G, uint2* cells,
                                  double* coarse, hipStream_t s)
{
    const size_t n = (size_t)G.nx * G.ny * G.nz;
    if (n == 0 || n_nodes <= 0) return hipSuccess;
    const int kx = (G.nx + 7) / 8, ky = (G.ny + 7) / 8, kz = (G.nz + 7) / 8;
    const size_t nc = (size_t)kx * ky * kz;
    hipLaunchKernelGGL(k_march_coarse, dim3((unsigned)((nc + 63) / 64)), dim3(64), 0, s, reinterpret_cast<const TriD<double>*>(tris_f64),
                       reinterpret_cast<const NodeD<double>*>(nodes_f64), n_nodes, G, kx, ky, kz, coarse);
    hipLaunchKernelGGL(k_march_fine, dim3((unsigned)((n + 255) / 256)), dim3(256), 0, s, reinterpret_cast<const TriD<double>*>(tris_f64),
                       reinterpret_cast<const NodeD<double>*>(nodes_f64), n_nodes, G, kx, ky, coarse, cells);
    return hipGetLastError();
}

// ---------------------------------------------------------------------------
// surface path tracer: trace_path + render_scene of S/path_tracing_fix1.py:18-169,
// one lane per pixel, samples looped in the lane (no atomics: a pixel has one owner)
// ---------------------------------------------------------------------------
LT_DEV void mark_unused(double* rand_0, size_t base, int from, int D)  // :36-38, :64-66, :128-130
{
    for (int b = from; b < D; b++) rand_0[base + b] = __builtin_huge_val();
}

// cast_one_shadow_ray (S/light_samples.py:36-61): radiance * brdf * geometry term * area of light sample `choice`
// as seen from X (offset along n); false when the sample is occluded.
__device__ __forceinline__ bool shadow_direct(const RenderParams& P, const TriD<double>* tris, const NodeD<double>* nodes, const int16_t* links,
                                              const double X[3], const double n[3], const lt_surface_material& M,
                                              int choice, double out[3])
{
    const double eps = 1e-6, inv_pi = 0.3183098861837907, inf = __builtin_huge_val();
    const double so[3] = {X[0] + eps * n[0], X[1] + eps * n[1], X[2] + eps * n[2]};
    const lt_point_light lt = P.lights[choice];
    double v[3] = {lt.source[0] - so[0], lt.source[1] - so[1], lt.source[2] - so[2]};
    const double mag = ::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
    const double sd[3] = {v[0] / mag, v[1] / mag, v[2] / mag};
    // (the reference takes the nearest hit of the shadow ray and asks whether it lies at mag - EPSILON or beyond, :49-52: the
    // sample is hidden exactly when SOME triangle is hit nearer than that, so the search stops at the first one it finds)
    int sp; double st;
    nearest_bvh_ordered(tris, nodes, P.n_nodes, links + ray_octant(sd) * 2 * P.n_nodes, so, sd, mag - eps, sp, st, true);
    (void)inf;
    if (sp >= 0) return false;
    const double cos_t = dot3(n, sd);
    const double nsd[3] = {-sd[0], -sd[1], -sd[2]};
    const double cos_p = dot3(lt.normal, nsd);
    const double geom = ::fabs(cos_t * cos_p) / (mag * mag);
#pragma unroll
    for (int k = 0; k < 3; k++) out[k] = (lt.radiance[k] * (M.diffuse[k] * inv_pi)) * geom * lt.total_area;
    return true;
}

// cos x for |x| <= 1 + rounding (the reference takes the cosine of a DOT PRODUCT of unit vectors, quirk B5): Taylor series to
// x^24 / 24!, truncation < 2e-26 -- OCML's general cos carries a Payne-Hanek reduction for huge arguments and its pow a log / exp pair with
// double-double arithmetic; inlined into the render kernels they cost ~100 VGPRs for one Schlick term
__device__ __forceinline__ double cos_unit(double x)
{
    const double z = x * x;
    double p = 1.6117373109360196e-24;                     //  1/24!
    p = __builtin_fma(p, z, -8.8967913924505741e-22);      // -1/22!
    p = __builtin_fma(p, z, 4.1103176233121648e-19);       //  1/20!
    p = __builtin_fma(p, z, -1.5619206968586225e-16);      // -1/18!
    p = __builtin_fma(p, z, 4.7794773323873853e-14);       //  1/16!
    p = __builtin_fma(p, z, -1.1470745597729725e-11);      // -1/14!
    p = __builtin_fma(p, z, 2.0876756987868099e-09);       //  1/12!
    p = __builtin_fma(p, z, -2.7557319223985891e-07);      // -1/10!
    p = __builtin_fma(p, z, 2.4801587301587302e-05);       //  1/8!
    p = __builtin_fma(p, z, -1.3888888888888889e-03);      // -1/6!
    p = __builtin_fma(p, z, 4.1666666666666664e-02);       //  1/4!
    p = __builtin_fma(p, z, -0.5);
    return __builtin_fma(p, z, 1.0);
}

// mirror (:82-85) and glass (:86-119, kept as written, quirk B5) branches of trace_path: new ray in o, d
__device__ __forceinline__ void specular_bounce(const lt_surface_material& M, const double X[3], const double n[3],
                                                bool inside, double r0, double o[3], double d[3])
{
    const double eps = 1e-6;
    if (!M.is_mirror) {
        const double n1 = inside ? M.ior : 1.0, n2 = inside ? 1.0 : M.ior;
        const double R0 = ((n1 - n2) / (n1 + n2)) * ((n1 - n2) / (n1 + n2));
        const double theta = dot3(d, n);
        // (theta is a dot product of unit vectors; x^5 by three products: within 2 ulp of pow(x, 5.0), the render's pins hold to 1e-9)
        const double om = 1 - cos_unit(theta), om2 = om * om;
        const double refl_prob = R0 + (1 - R0) * (om2 * om2 * om);
        double Nr = M.ior;
        if (theta > 0) Nr = 1 / Nr;
        Nr = 1 / Nr;
        const double cos_theta = -theta;
        const double rad = 1 - (Nr * Nr) * (1 - cos_theta * cos_theta);
        if (rad > 0 && r0 > refl_prob) {
            const double kk = Nr * cos_theta - ::sqrt(rad);
            double tr[3] = {d[0] * Nr + n[0] * kk, d[1] * Nr + n[1] * kk, d[2] * Nr + n[2] * kk};
            normalize3(tr);
#pragma unroll
            for (int k = 0; k < 3; k++) { o[k] = X[k] - eps * n[k]; d[k] = tr[k]; }
            return;
        }
    }
    double r[3]; reflect(d, n, r);
#pragma unroll
    for (int k = 0; k < 3; k++) { o[k] = X[k] + eps * n[k]; d[k] = r[k]; }
}


// One lane per PATH (pixel, sample), 256 lanes per workgroup: a workgroup owns PPB = max(1, 256 / S) whole pixels, its lanes
// trace their paths side by side, park the radiance in LDS, and one lane per pixel then adds its samples in ascending order --
// the reference's own order of summation (:148-164), so the image is what the sample loop of round 1-3's kernel gave, bit for
// bit.  (That kernel ran one lane per pixel over all its samples: 90 000 lanes for the notebook's 300 x 300 render, a third of
// the device's lanes for 50 sequential paths each.)  Pixels with more than 256 samples take several rounds.  The scene tables
// (triangles, BVH nodes, materials: 7 KB for the notebook's scene) are staged in LDS when they fit 48 KiB (LDS_TABLES); the
// 2000 point lights (160 KB) and the random tables stay in global memory.
constexpr int kRenderThreads = 256;
#ifndef LT_RENDER_WAVES
#define LT_RENDER_WAVES 4      // waves per SIMD the register allocator leaves room for: 2 (172 VGPRs, no scratch) 7.1 ms, 3 (168 + 12 B)
                               // 5.2 ms, 4 (128 + 188 B of scratch) 4.6 ms for the 300 x 300 x 50 spp render (profiles/r04e_render_time.log)
#endif
template <bool LDS_TABLES>
__global__ void __launch_bounds__(kRenderThreads, LT_RENDER_WAVES) k_render_surface(const RenderParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char r_lds[];
    double* s_L = reinterpret_cast<double*>(r_lds);                                       // [256][3] radiance of the lanes' paths
    const size_t off_t = (size_t)kRenderThreads * 3 * sizeof(double);
    const size_t off_n = off_t + (size_t)P.n_tris * sizeof(TriD<double>), off_m = off_n + (size_t)P.n_nodes * sizeof(NodeD<double>);
    const TriD<double>* tris = LDS_TABLES ? reinterpret_cast<const TriD<double>*>(r_lds + off_t) : reinterpret_cast<const TriD<double>*>(P.tris);
    const NodeD<double>* nodes = LDS_TABLES ? reinterpret_cast<const NodeD<double>*>(r_lds + off_n) : reinterpret_cast<const NodeD<double>*>(P.nodes);
    const lt_surface_material* mats = LDS_TABLES ? reinterpret_cast<const lt_surface_material*>(r_lds + off_m) : P.mats;
    const size_t off_k = off_m + (size_t)P.n_tris * sizeof(lt_surface_material);
    const int16_t* links = LDS_TABLES ? reinterpret_cast<const int16_t*>(r_lds + off_k) : P.links;
    if constexpr (LDS_TABLES) {
        lds_copy(r_lds + off_k, P.links, (size_t)((16 * P.n_nodes * 2 + 3) / 4) * 4);
        lds_copy(r_lds + off_t, P.tris, (size_t)P.n_tris * sizeof(TriD<double>));
        lds_copy(r_lds + off_n, P.nodes, (size_t)P.n_nodes * sizeof(NodeD<double>));
        lds_copy(r_lds + off_m, P.mats, (size_t)P.n_tris * sizeof(lt_surface_material));
        __syncthreads();
    }
    const double eps = 1e-6, inv_pi = 0.3183098861837907, inf = __builtin_huge_val();
    const int tid = threadIdx.x;
    const int chunk = P.S < kRenderThreads ? P.S : kRenderThreads;       // samples of one pixel traced per round
    const int ppb = kRenderThreads / chunk;                              // whole pixels per workgroup
    const int pl = tid / chunk, sl = tid % chunk;                        // this lane's pixel (local) and sample slot
    const int pix = blockIdx.x * ppb + pl;
    const bool owner = tid < ppb && blockIdx.x * ppb + tid < P.W * P.H;  // lane tid sums local pixel tid
    double color[3] = {0, 0, 0};
    for (int s0 = 0; s0 < P.S; s0 += chunk) {
        const int smp = s0 + sl;
        double L[3] = {0, 0, 0};
        if (pl < ppb && pix < P.W * P.H && smp < P.S) {
            const int i = pix / P.W, j = pix % P.W;
            const size_t base = (((size_t)i * P.W + j) * P.S + smp) * (size_t)P.D;
            // camera ray, :152-160 (anti-alias jitter re-uses the bounce-0 uniform for x and y)
            double o[3] = {P.cam[0], P.cam[1], P.cam[2]};
            const double jit = P.rand_0[base];
            double d[3] = {P.xs[j] + jit / (double)P.W - o[0], P.ys[i] + jit / (double)P.H - o[1], P.f_distance - o[2]};
            normalize3(d);
            double thr[3] = {1, 1, 1};
            // ONE traversal site serves both kinds of ray: (o, d) is the ray the next search follows -- the path's own ray, or,
            // while `shadow`, the shadow ray of the diffuse hit (`hit`, `thit` on the stashed path ray po, pd) whose shading
            // is completed when the search returns.  Two inlined copies of the traversal cost the kernel 221 VGPRs; the
            // expressions evaluated are those of trace_path / cast_one_shadow_ray, in the same order.
            bool shadow = false;
            int hit = -1; double thit = 0, smag = 0;
            double po[3] = {0, 0, 0}, pd[3] = {0, 0, 1};
            for (int bounce = 0;;) {
                if (!shadow && bounce >= P.D) break;                        // :24-26
                int prim; double t;
                nearest_bvh_ordered(tris, nodes, P.n_nodes, links + ray_octant(d) * 2 * P.n_nodes, o, d, shadow ? smag - eps : inf, prim, t, shadow);    // hit_object, :32 / cast_one_shadow_ray
                const double r0 = P.rand_0[base + bounce], r1 = P.rand_1[base + bounce];
                if (!shadow) {
                    if (prim < 0) { mark_unused(P.rand_0, base, bounce, P.D); break; }
                    const lt_surface_material M = mats[prim];
                    double n[3] = {tris[prim].n[0], tris[prim].n[1], tris[prim].n[2]};
                    const double X[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
                    if (M.is_light) { L[0] += M.emission * thr[0]; L[1] += M.emission * thr[1]; L[2] += M.emission * thr[2]; }
                    bool inside = false;
                    if (dot3(n, d) > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; inside = true; }   // :48-51
                    if (M.is_diffuse) {
                        // direct light: cast_one_shadow_ray, S/light_samples.py:36-61 -- the shadow ray becomes the ray to follow
                        const lt_point_light lt = P.lights[P.light_choice[base + bounce]];
#pragma unroll
                        for (int k = 0; k < 3; k++) { po[k] = o[k]; pd[k] = d[k]; o[k] = X[k] + eps * n[k]; }
                        double v[3] = {lt.source[0] - o[0], lt.source[1] - o[1], lt.source[2] - o[2]};
                        smag = ::sqrt(v[0] * v[0] + v[1] * v[1] + v[2] * v[2]);
                        d[0] = v[0] / smag; d[1] = v[1] / smag; d[2] = v[2] / smag;
                        hit = prim; thit = t; shadow = true;
                        continue;
                    } else if (M.is_mirror || M.transmission > 0.0) {           // :82-119
                        specular_bounce(M, X, n, inside, r0, o, d);
                    } else break;                                               // :121-123
                } else {
                    // the shadow ray is back: finish the diffuse hit it belongs to
                    shadow = false;
                    const lt_surface_material M = mats[hit];
                    double n[3] = {tris[hit].n[0], tris[hit].n[1], tris[hit].n[2]};
                    if (dot3(n, pd) > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; }
                    const double X[3] = {po[0] + thit * pd[0], po[1] + thit * pd[1], po[2] + thit * pd[2]};
                    if (prim < 0) {      // nothing nearer than the light sample: radiance * brdf * geometry term * area, :53-61
                        const lt_point_light lt = P.lights[P.light_choice[base + bounce]];
                        const double cos_t = dot3(n, d);
                        const double nsd[3] = {-d[0], -d[1], -d[2]};
                        const double cos_p = dot3(lt.normal, nsd);
                        const double geom = ::fabs(cos_t * cos_p) / (smag * smag);
#pragma unroll
                        for (int k = 0; k < 3; k++) L[k] += thr[k] * ((lt.radiance[k] * (M.diffuse[k] * inv_pi)) * geom * lt.total_area);
                    }
                    // indirect: cosine lobe, :63-80
                    double o4[4];
                    cosine_hemi<double, true>(n, pd, r0, r1, o4);
                    if (o4[3] == 0) { mark_unused(P.rand_0, base, bounce + 1, P.D); break; }
                    const double cos_theta = o4[0] * n[0] + o4[1] * n[1] + o4[2] * n[2];
#pragma unroll
                    for (int k = 0; k < 3; k++) {
                        thr[k] *= (M.diffuse[k] * inv_pi) * cos_theta / o4[3];
                        o[k] = X[k] + eps * o4[k];
                        d[k] = o4[k];
                    }
                }
                if (bounce > 5) {                                           // russian roulette, :126-132
                    const double rr = ::fmax(0.05, 1 - thr[1]);
                    if (r0 < rr) { mark_unused(P.rand_0, base, bounce + 1, P.D); break; }
                    thr[0] /= 1 - rr; thr[1] /= 1 - rr; thr[2] /= 1 - rr;
                }
                bounce++;
            }
        }
        s_L[tid * 3] = L[0]; s_L[tid * 3 + 1] = L[1]; s_L[tid * 3 + 2] = L[2];
        __syncthreads();
        if (owner) {        // color += light, sample by sample (:162)
            const int ns = P.S - s0 < chunk ? P.S - s0 : chunk;
            for (int q = 0; q < ns; q++) {
                const double* l = s_L + (size_t)(tid * chunk + q) * 3;
                color[0] += l[0]; color[1] += l[1]; color[2] += l[2];
            }
        }
        __syncthreads();
    }
    if (owner) {
        const size_t px = (size_t)blockIdx.x * ppb + tid;
#pragma unroll
        for (int k = 0; k < 3; k++) {                                       // :164-166
            double c = color[k] / (double)P.S;
            c = c < 0 ? 0.0 : (c > 1 ? 1.0 : c);
            P.image[px * 3 + k] += 0.25 * c;
        }
    }
}

// path_tracing_old.py:17-171 -- the recursive ancestor of trace_path that examples/LTS.ipynb still calls.  A diffuse hit
// calls trace_path(ray, bounce + 1) on the SHARED ray and, once that returns, carries on from wherever the callee left
// the ray (:68-80); emission counts at bounce 0 only (:45), the direct term is not throughput-weighted (:80), roulette
// starts after bounce 3 (:127) and the pixel is overwritten with clip(mean) (:167).  The recursion is unrolled into an
// explicit per-lane stack of (throughput, light, direct, r0, bounce) frames, visited in the reference's depth-first
// order so that the +inf markers written into rand_0 are read back exactly where the reference reads them.
// One lane per PATH (pixel, sample) as in k_render_surface, one wave per workgroup; the frame being executed lives in
// registers, the frames of its callers in LDS ([depth][field][lane]: at most D of them, 45 KiB for the notebook's depth 8) --
// rounds 1-3 kept the whole stack of 25 frames in scratch memory (2288 B per lane) and looped over a pixel's samples in one lane.
constexpr int kOldThreads = 64, kOldFields = 11;      // thr[3], L[3], direct[3], r0, bounce
__global__ void __launch_bounds__(kOldThreads) k_render_surface_old(const RenderParams P)
{
    extern __shared__ __attribute__((aligned(16))) unsigned char r_lds[];
    double* s_L = reinterpret_cast<double*>(r_lds);                       // [64][3]
    double* s_st = s_L + kOldThreads * 3;                                 // [D][kOldFields][64]
    const TriD<double>* tris = reinterpret_cast<const TriD<double>*>(P.tris);
    const NodeD<double>* nodes = reinterpret_cast<const NodeD<double>*>(P.nodes);
    const double eps = 1e-6, inv_pi = 0.3183098861837907, inf = __builtin_huge_val();
    const int tid = threadIdx.x;
    const int chunk = P.S < kOldThreads ? P.S : kOldThreads, ppb = kOldThreads / chunk;
    const int pl = tid / chunk, sl = tid % chunk;
    const int pix = blockIdx.x * ppb + pl;
    const bool owner = tid < ppb && blockIdx.x * ppb + tid < P.W * P.H;
    double color[3] = {0, 0, 0};
    for (int s0 = 0; s0 < P.S; s0 += chunk) {
        const int smp = s0 + sl;
        double ret[3] = {0, 0, 0};
        if (pl < ppb && pix < P.W * P.H && smp < P.S) {
            const int i = pix / P.W, j = pix % P.W;
            const size_t sample = ((size_t)i * P.W + j) * P.S + smp;
            const size_t base = sample * (size_t)P.D;
            const int32_t* choice = P.light_choice + sample * (size_t)P.choices;
            unsigned n_shadow = 0;
            double o[3] = {P.cam[0], P.cam[1], P.cam[2]};
            const double jit = P.rand_0[base];                               // :158-159
            double d[3] = {P.xs[j] + jit / (double)P.W - o[0], P.ys[i] + jit / (double)P.H - o[1], P.f_distance - o[2]};
            normalize3(d);
            int depth = 0;
            bool resume = false;
            // the frame in execution
            double f_thr[3] = {1, 1, 1}, f_L[3] = {0, 0, 0}, f_direct[3] = {0, 0, 0}, f_r0 = 0.0;
            int f_bounce = 0;
            for (;;) {
                bool done = false;
                if (resume) {                                                // back from trace_path(bounce + 1), :78-80
                    resume = false;
#pragma unroll
                    for (int k = 0; k < 3; k++) f_L[k] += (f_direct[k] + f_thr[k] * ret[k]);
                } else if (f_bounce >= P.D) {                                // :24-25
                    done = true;
                } else {
                    const double r0 = P.rand_0[base + f_bounce], r1 = P.rand_1[base + f_bounce];
                    f_r0 = r0;
                    int prim; double t;
                    nearest_bvh_ordered(tris, nodes, P.n_nodes, P.links + ray_octant(d) * 2 * P.n_nodes, o, d, inf, prim, t);
                    if (prim < 0) { mark_unused(P.rand_0, base, f_bounce, P.D); done = true; }
                    else {
                        const lt_surface_material M = P.mats[prim];
                        double n[3] = {tris[prim].n[0], tris[prim].n[1], tris[prim].n[2]};
                        const double X[3] = {o[0] + t * d[0], o[1] + t * d[1], o[2] + t * d[2]};
                        if (M.is_light && f_bounce == 0) {                   // :45-46
#pragma unroll
                            for (int k = 0; k < 3; k++) f_L[k] += M.emission * f_thr[k];
                        }
                        bool inside = false;
                        if (dot3(n, d) > 0) { n[0] = -n[0]; n[1] = -n[1]; n[2] = -n[2]; inside = true; }
                        if (M.is_diffuse) {
                            if (!shadow_direct(P, tris, nodes, P.links, X, n, M, choice[n_shadow % (unsigned)P.choices], f_direct))
                                f_direct[0] = f_direct[1] = f_direct[2] = 0;
                            n_shadow++;
                            double o4[4];
                            cosine_hemi<double, true>(n, d, r0, r1, o4);
                            if (o4[3] == 0) { mark_unused(P.rand_0, base, f_bounce + 1, P.D); done = true; }
                            else {
                                const double cos_theta = o4[0] * n[0] + o4[1] * n[1] + o4[2] * n[2];
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    f_thr[k] *= (M.diffuse[k] * inv_pi) * cos_theta / o4[3];
                                    o[k] = X[k] + eps * o4[k];
                                    d[k] = o4[k];
                                }
                                // trace_path(scene, ..., ray, bounce + 1, ...), :78 -- the caller's frame goes to LDS
                                double* fr = s_st + ((size_t)depth * kOldFields) * kOldThreads + tid;
#pragma unroll
                                for (int k = 0; k < 3; k++) {
                                    fr[(size_t)k * kOldThreads] = f_thr[k]; fr[(size_t)(3 + k) * kOldThreads] = f_L[k];
                                    fr[(size_t)(6 + k) * kOldThreads] = f_direct[k];
                                    f_thr[k] = 1; f_L[k] = 0; f_direct[k] = 0;
                                }
                                fr[(size_t)9 * kOldThreads] = f_r0; fr[(size_t)10 * kOldThreads] = (double)f_bounce;
                                f_r0 = 0.0; f_bounce = f_bounce + 1;
                                depth++;
                                continue;
                            }
                        } else if (M.is_mirror || M.transmission > 0.0) {
                            specular_bounce(M, X, n, inside, r0, o, d);
                        } else done = true;                                  // :122-124
                    }
                }
                if (!done && f_bounce > 3) {                                 // :127-133
                    const double rr = ::fmax(0.05, 1 - f_thr[1]);
                    if (f_r0 < rr) { mark_unused(P.rand_0, base, f_bounce + 1, P.D); done = true; }
                    else { f_thr[0] /= 1 - rr; f_thr[1] /= 1 - rr; f_thr[2] /= 1 - rr; }
                }
                if (!done) { f_bounce++; continue; }
                ret[0] = f_L[0]; ret[1] = f_L[1]; ret[2] = f_L[2];          // return light, :137
                if (depth == 0) break;
                depth--;                                                     // the caller's frame comes back
                const double* fr = s_st + ((size_t)depth * kOldFields) * kOldThreads + tid;
#pragma unroll
                for (int k = 0; k < 3; k++) {
                    f_thr[k] = fr[(size_t)k * kOldThreads]; f_L[k] = fr[(size_t)(3 + k) * kOldThreads];
                    f_direct[k] = fr[(size_t)(6 + k) * kOldThreads];
                }
                f_r0 = fr[(size_t)9 * kOldThreads]; f_bounce = (int)fr[(size_t)10 * kOldThreads];
                resume = true;
            }
        }
        s_L[tid * 3] = ret[0]; s_L[tid * 3 + 1] = ret[1]; s_L[tid * 3 + 2] = ret[2];
        __syncthreads();
        if (owner) {        // color += light, sample by sample (:163)
            const int ns = P.S - s0 < chunk ? P.S - s0 : chunk;
            for (int q = 0; q < ns; q++) {
                const double* l = s_L + (size_t)(tid * chunk + q) * 3;
                color[0] += l[0]; color[1] += l[1]; color[2] += l[2];
            }
        }
        __syncthreads();
    }
    if (owner) {
        const size_t px = (size_t)blockIdx.x * ppb + tid;
#pragma unroll
        for (int k = 0; k < 3; k++) {                                       // :166-167
            double c = color[k] / (double)P.S;
            c = c < 0 ? 0.0 : (c > 1 ? 1.0 : c);
            P.image[px * 3 + k] = c;
        }
    }
}

hipError_t launch_render_surface(const RenderParams& P, hipStream_t s)
{
    const int n = P.W * P.H;
    if (n <= 0) return hipSuccess;
    if (P.variant == 1) {
        const int chunk_o = P.S < kOldThreads ? P.S : kOldThreads, ppb_o = kOldThreads / chunk_o;
        const size_t lds_o = ((size_t)kOldThreads * 3 + (size_t)P.D * kOldFields * kOldThreads) * sizeof(double);      // D <= kRenderOldMaxDepth: <= 137 KiB
        if (lds_o > 48 * 1024) {
            hipError_t e = hipFuncSetAttribute(reinterpret_cast<const void*>(&k_render_surface_old), hipFuncAttributeMaxDynamicSharedMemorySize, (int)lds_o);
            if (e != hipSuccess) return e;
        }
        hipLaunchKernelGGL(k_render_surface_old, dim3((n + ppb_o - 1) / ppb_o), dim3(kOldThreads), lds_o, s, P);
        return hipGetLastError();
    }
    const int chunk = P.S < kRenderThreads ? P.S : kRenderThreads, ppb = kRenderThreads / chunk;
    const size_t tables = (size_t)P.n_tris * (sizeof(TriD<double>) + sizeof(lt_surface_material)) + (size_t)P.n_nodes * sizeof(NodeD<double>) +
                          (size_t)((16 * P.n_nodes * 2 + 3) / 4) * 4;
    const size_t lds_l = (size_t)kRenderThreads * 3 * sizeof(double);
    if (tables <= 48 * 1024)
        hipLaunchKernelGGL(k_render_surface<true>, dim3((n + ppb - 1) / ppb), dim3(kRenderThreads), lds_l + tables, s, P);
    else
        hipLaunchKernelGGL(k_render_surface<false>, dim3((n + ppb - 1) / ppb), dim3(kRenderThreads), lds_l, s, P);
    return hipGetLastError();
}

static inline unsigned nblk(size_t n, unsigned t) { return (unsigned)((n + t - 1) / t); }

hipError_t launch_intersect_rays(const void* tris, const void* nodes, int n_tris, int n_nodes, const double* o,
                                 const double* d, const double* tmax, size_t n, int use_bvh, const MarchGrid* G, const int16_t* links,
                                 int32_t* prim, double* t, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    MarchGrid g;
    if (use_bvh == 4 && !links) use_bvh = 1;
    if (G) g = *G; else { memset(&g, 0, sizeof g); if (use_bvh == 2 || use_bvh == 3) use_bvh = 1; }
    hipLaunchKernelGGL(k_intersect_rays, dim3(nblk(n, 256)), dim3(256), 0, s,
                       reinterpret_cast<const TriD<double>*>(tris), reinterpret_cast<const NodeD<double>*>(nodes),
                       n_tris, n_nodes, o, d, tmax, n, use_bvh, g, links, prim, t);
    return hipGetLastError();
}
hipError_t launch_triangle_intersect(const double* o, const double* d, const double* tris, size_t n, double* t,
                                     hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_triangle_intersect, dim3(nblk(n, 256)), dim3(256), 0, s, o, d, tris, n, t);
    return hipGetLastError();
}
hipError_t launch_intersect_bounds(const double* o, const double* d, const double* tmax, const double* boxes,
                                   size_t n, int32_t* hit, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_intersect_bounds, dim3(nblk(n, 256)), dim3(256), 0, s, o, d, tmax, boxes, n, hit);
    return hipGetLastError();
}
hipError_t launch_eval(int fn, const double* in, size_t n, double* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_eval, dim3(nblk(n, 256)), dim3(256), 0, s, fn, in, n, out);
    return hipGetLastError();
}
hipError_t launch_rng_raw(unsigned long long seed, unsigned long long photon_id, unsigned count, uint32_t* out,
                          hipStream_t s)
{
    hipLaunchKernelGGL(k_rng_raw, dim3(1), dim3(64), 0, s, seed, photon_id, count, out);
    return hipGetLastError();
}
hipError_t launch_grid_to_f64(const void* grid, int tally, size_t n, double* out, hipStream_t s)
{
    if (n == 0) return hipSuccess;
    hipLaunchKernelGGL(k_grid_to_f64, dim3(2048), dim3(256), 0, s, grid, tally, n, out);
    return hipGetLastError();
}

hipError_t launch_grid_add(void* dst, const void* src, int tally, size_t n, hipStream_t s)
{
    if (tally == LT_TALLY_F32) hipLaunchKernelGGL(k_grid_add<float>, dim3(2048), dim3(256), 0, s, (float*)dst, (const float*)src, n);
    else if (tally == LT_TALLY_F64) hipLaunchKernelGGL(k_grid_add<double>, dim3(2048), dim3(256), 0, s, (double*)dst, (const double*)src, n);
    else hipLaunchKernelGGL(k_grid_add<unsigned long long>, dim3(2048), dim3(256), 0, s, (unsigned long long*)dst, (const unsigned long long*)src, n);
    return hipGetLastError();
}

}  // namespace ltk
