/*
** oracle/ref_headers/upwelling_my25.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure): the options of upwelling_nodiag.h
** (= the numerical options of ROMS/Include/upwelling.h) with ANA_VMIX replaced by the
** Mellor-Yamada level 2.5 closure and the options ROMS applications give it
** (MY25_MIXING + KANTHA_CLAYSON + N2S2_HORAVG + RI_SPLINES), so that my25_prestep.F,
** my25_corstep.F and tkebc_im.F are built.  An application header is user
** configuration in ROMS (cppdefs.h:655-668).
*/
#define UV_ADV
#define UV_COR
#define UV_LDRAG
#define UV_VIS2
#define MIX_S_UV
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define DJ_GRADPS
#define TS_DIF2
#define MIX_S_TS
#define SALINITY
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
#define MY25_MIXING
#define KANTHA_CLAYSON
#define N2S2_HORAVG
#define RI_SPLINES
