// One translation unit per (curve, group): compile with -DGMSM_GROUP_ID=<enum gmsm_group value>.
#include "gmsm_group.h"

namespace gmsm {

#if GMSM_GROUP_ID == 0
using TheGroup = Group<Fp<bn254_fp_params>, bn254_fr_params, bn254_g1_consts, false>;  // prime order: no torsion check
#define GMSM_VT_NAME gmsm_vtable_bn254_g1
#elif GMSM_GROUP_ID == 1
using TheGroup = Group<Fp2<bn254_fp_params>, bn254_fr_params, bn254_g2_consts, true>;
#define GMSM_VT_NAME gmsm_vtable_bn254_g2
#elif GMSM_GROUP_ID == 2
using TheGroup = Group<Fp<bls12_381_fp_params>, bls12_381_fr_params, bls12_381_g1_consts, true>;
#define GMSM_VT_NAME gmsm_vtable_bls12_381_g1
#elif GMSM_GROUP_ID == 3
using TheGroup = Group<Fp2<bls12_381_fp_params>, bls12_381_fr_params, bls12_381_g2_consts, true>;
#define GMSM_VT_NAME gmsm_vtable_bls12_381_g2
#elif GMSM_GROUP_ID == 4
using TheGroup = Group<Fp<bw6_761_fp_params>, bw6_761_fr_params, bw6_761_g1_consts, true>;
#define GMSM_VT_NAME gmsm_vtable_bw6_761_g1
#elif GMSM_GROUP_ID == 5
// BW6-761 G2 is over Fp as well (ecc/bw6-761/g2.go:20); a distinct tag type keeps its kernels in their own TU
struct bw6_761_g2_fr_params : bw6_761_fr_params {};
using TheGroup = Group<Fp<bw6_761_fp_params>, bw6_761_g2_fr_params, bw6_761_g2_consts, true>;
#define GMSM_VT_NAME gmsm_vtable_bw6_761_g2
#else
#error "GMSM_GROUP_ID must be 0..5"
#endif

const GroupVTable *GMSM_VT_NAME() { return VTableOf<TheGroup>::get(); }

}  // namespace gmsm
