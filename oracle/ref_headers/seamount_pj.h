/*
** oracle/ref_headers/seamount_pj.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure): the options of seamount_nodiag.h
** (= the numerical options of ROMS/Include/seamount.h) with DJ_GRADPS replaced
** by PJ_GRADP,
** so that prsgrd.F selects the finite-volume pressure Jacobian prsgrd40.h (prsgrd.F:20-21).
** An application header is user configuration in ROMS (cppdefs.h:655-668).
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_S_UV
#define PJ_GRADP
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
