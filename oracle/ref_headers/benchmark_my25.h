/*
** oracle/ref_headers/benchmark_my25.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure): the options of ROMS/Include/benchmark.h
** with the KPP block (LMD_MIXING ...) replaced by MY25_MIXING alone (no KANTHA_CLAYSON, no RI_SPLINES,
** no N2S2_HORAVG: the plain shear and the Galperin et al. stability functions of my25_corstep.F).
** An application header is user configuration in ROMS (cppdefs.h:655-668).
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define MIX_S_UV
#define DJ_GRADPS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define TS_DIF2
#define MIX_GEO_TS
#define SOLAR_SOURCE
#define NONLIN_EOS
#define SALINITY
#define CURVGRID
#define SOLVE3D
#define MY25_MIXING
#define BULK_FLUXES
#ifdef BULK_FLUXES
# define ANA_WINDS
# define ANA_TAIR
# define ANA_PAIR
# define ANA_HUMIDITY
# define ANA_RAIN
# define LONGWAVE
# define ANA_CLOUD
#endif
#define SPHERICAL
#define ANA_GRID
#define ANA_INITIAL
#define ALBEDO
#define ANA_SRFLUX
#define ANA_SSFLUX
#define ANA_BSFLUX
#define ANA_BTFLUX
