/*
** oracle/ref_headers/seamount_geouv.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure): the options of seamount_nodiag.h with
** MIX_S_UV replaced by MIX_GEO_UV, so that uv3dmix.F includes uv3dmix2_geo.h (the harmonic
** viscosity rotated to geopotential surfaces).  An application header is user
** configuration in ROMS (cppdefs.h:655-668).
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_GEO_UV
#define DJ_GRADPS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define TS_DIF2
#define MIX_GEO_TS
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_BTFLUX
