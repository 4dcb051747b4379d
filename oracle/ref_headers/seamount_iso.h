/*
** oracle/ref_headers/seamount_iso.h -- application option list used ONLY by
** oracle/build_ref.sh (test infrastructure).  It selects, for the reference
** files compiled there, the numerical options of the SEAMOUNT test case
** (ROMS/Include/seamount.h:15-31) WITHOUT ANA_DIAG: that option only adds the
** user diagnostics of Functionals/ana_diag.h, which does not compile under
** IMPLICIT NONE (it declares io_error and uses io_err, ana_diag.h:96/114) and
** would keep analytical.F and diag.F out of the build.  An application header
** is user configuration in ROMS (cppdefs.h:655-668).
** This variant adds TS_DIF4 and UV_VIS4 and replaces MIX_GEO_TS by MIX_ISO_TS: t3dmix2_iso.h, t3dmix4_iso.h.
*/
#define UV_ADV
#define UV_COR
#define UV_QDRAG
#define UV_VIS2
#define UV_VIS4
#define MIX_S_UV
#define DJ_GRADPS
#define SPLINES_VDIFF
#define SPLINES_VVISC
#define TS_DIF2
#define TS_DIF4
#define MIX_ISO_TS
#define SOLVE3D
#define ANA_GRID
#define ANA_INITIAL
#define ANA_SMFLUX
#define ANA_STFLUX
#define ANA_BTFLUX
