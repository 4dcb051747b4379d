/*
** oracle/ref_headers/upwelling_nodiag.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure).  It selects, for the reference
** files compiled there, the numerical options of the UPWELLING test case
** (ROMS/Include/upwelling.h:15-49) WITHOUT the output-side options AVERAGES,
** DIAGNOSTICS_TS, DIAGNOSTICS_UV and PERFECT_RESTART, whose bookkeeping arrays
** need the I/O metadata (varinfo) layer that is not part of this build.  An
** application header is user configuration in ROMS (cppdefs.h:655-668).
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
#define ANA_VMIX
