/*
** oracle/ref_headers/upwelling_iso.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure): the options of upwelling_nodiag.h
** (= the numerical options of ROMS/Include/upwelling.h) plus the biharmonic
** operators TS_DIF4 and UV_VIS4 (t3dmix.F and uv3dmix.F then build t3dmix4_s.h and
** uv3dmix4_s.h beside the harmonic ones), and MIX_ISO_TS in the place of MIX_S_TS: the tracer
** operators are then t3dmix2_iso.h and t3dmix4_iso.h.  An application header is user
** configuration in ROMS (cppdefs.h:655-668).
*/
#define UV_ADV
#define UV_COR
#define UV_LDRAG
#define UV_VIS2
#define UV_VIS4
#define MIX_S_UV
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define DJ_GRADPS
#define TS_DIF2
#define TS_DIF4
#define MIX_ISO_TS
#define SALINITY
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_SSFLUX
#define ANA_BTFLUX
#define ANA_BSFLUX
#define ANA_VMIX
