#!/bin/bash
# oracle/build_ref.sh -- TEST INFRASTRUCTURE ONLY.
#
# Builds, from the reference sources WHERE THEY LIE under /root/reference, the
# subset of the hot-path files that compiles stand-alone with flang -- i.e.
# without netCDF-Fortran, which the image lacks (no stand-ins are written for
# it; files that need it, directly or through mod_sources -> mod_netcdf, are
# simply not part of this build: step2d, pre_step3d, rhs3d, step3d_uv,
# step3d_t, omega).  Outputs go only to oracle/_ref/ (git-ignored):
#
#   oracle/_ref/<APP>/libref.so   reference objects + our bind(C) wrapper
#                                 (oracle/ref_wrap.F90) for APP in
#                                 BENCHMARK, UPWELLING, SEAMOUNT; BENCHMARK_MASK, UPWELLING_MASK
#                                 (= the application + -DMASKING); UPWELLING_PG31, UPWELLING_WJ,
#                                 SEAMOUNT_PG31, SEAMOUNT_WJ (= the application with prsgrd31.h, plain
#                                 and with WJ_GRADP, instead of prsgrd32.h); UPWELLING_DIF4, SEAMOUNT_DIF4
#                                 (= the application plus TS_DIF4 and UV_VIS4), UPWELLING_MASK_DIF4; UPWELLING_ISO, SEAMOUNT_ISO,
#                                 UPWELLING_MASK_ISO (= the _DIF4 options with MIX_ISO_TS as the tracer mixing choice); UPWELLING_LOGDRAG
#                                 (UV_LOGDRAG instead of UV_LDRAG); UPWELLING_PJ, SEAMOUNT_PJ (PJ_GRADP: prsgrd40.h); UPWELLING_RAD2D,
#                                 UPWELLING_MASK_RAD2D, BENCHMARK_RAD2D (+ -DRADIATION_2D); UPWELLING_GLS, UPWELLING_MASK_GLS
#                                 (GLS_MIXING + KANTHA_CLAYSON + N2S2_HORAVG + RI_SPLINES instead of ANA_VMIX), BENCHMARK_GLS
#                                 (GLS_MIXING + CANUTO_A instead of the KPP block): gls_prestep.F, gls_corstep.F, tkebc_im.F;
#                                 UPWELLING_MY25, UPWELLING_MASK_MY25, BENCHMARK_MY25 (MY25_MIXING: my25_prestep.F, my25_corstep.F);
#                                 UPWELLING_GEOUV, SEAMOUNT_GEOUV, UPWELLING_MASK_GEOUV, UPWELLING_MASK_WET_GEOUV (MIX_GEO_UV instead of
#                                 MIX_S_UV: uv3dmix2_geo.h);
#                                 UPWELLING_ATM[_PG31|_PJ] (+ -DATM_PRESS: the air-pressure term of the three pressure-gradient files);
#                                 UPWELLING_MASK_WET[_DIF4|_ISO|_PG31], BENCHMARK_MASK_WET (+ -DWET_DRY; PJ_GRADP with WET_DRY does not
#                                 compile in the reference itself: prsgrd40.h:98-100 passes umask_wet, vmask_wet without declaring them);
#                                 UPWELLING_STAB_DIF4, SEAMOUNT_STAB_DIF4, UPWELLING_STAB_ISO, SEAMOUNT_STAB_ISO (+ -DTS_MIX_STABILITY);
#                                 UPWELLING_MINSTRAT_ISO, SEAMOUNT_MINSTRAT_ISO (+ -DTS_MIX_MIN_STRAT)
#
# This is the reference's own recipe (makefile:207, Compilers/Linux-gfortran.mk:
# 43-44: cpp -P -traditional then the Fortran compiler), serial build (no
# DISTRIBUTE/MPI), flags -O2 without fast-math and without FMA contraction.
set -e
REF=${REF:-/root/reference}
HERE=$(cd "$(dirname "$0")" && pwd)
OUT=$HERE/_ref
FC=${FC:-flang}
FFLAGS="-O2 -fPIC -ffp-contract=off"
[ -d "$REF/ROMS" ] || { echo "reference not present: nothing to build"; exit 0; }
command -v $FC >/dev/null || { echo "flang not present: nothing to build"; exit 0; }

# dependency-ordered list (reference files only)
FILES="Modules/mod_kinds Modules/mod_param Modules/mod_strings Modules/mod_iounits Modules/mod_scalars
 Modules/mod_parallel Utility/strings Utility/yaml_parser Utility/get_metadata Modules/mod_ncparam
 Modules/mod_grid Modules/mod_ocean Modules/mod_coupling Modules/mod_mixing Modules/mod_forces
 Modules/mod_stepping Modules/mod_clima Modules/mod_boundary Modules/mod_eoscoef Modules/mod_diags
 Utility/round Utility/dateclock Nonlinear/exchange_2d Nonlinear/exchange_3d Utility/get_bounds
 Utility/set_weights Utility/mp_routines Utility/timers Nonlinear/prsgrd Nonlinear/t3dmix Nonlinear/uv3dmix Nonlinear/set_depth
  Nonlinear/set_massflux Nonlinear/rho_eos Nonlinear/set_zeta Nonlinear/mpdata_adiff
 Nonlinear/bc_2d Nonlinear/set_vbc Nonlinear/bulk_flux
 Nonlinear/zetabc Nonlinear/u2dbc_im Nonlinear/v2dbc_im Nonlinear/u3dbc_im Nonlinear/v3dbc_im Nonlinear/t3dbc_im
 Nonlinear/ini_fields
 Nonlinear/bc_3d Utility/shapiro Nonlinear/lmd_swfrac Nonlinear/lmd_skpp Nonlinear/lmd_vmix
 Nonlinear/tkebc_im Nonlinear/gls_prestep Nonlinear/gls_corstep Nonlinear/my25_prestep Nonlinear/my25_corstep

 Utility/stats Functionals/analytical Nonlinear/wvelocity Nonlinear/diag Utility/set_scoord Utility/metrics"

build_app () {
  # <APP>_MASK: the same application with the MASKING option added on the command line (a CPP option of the
  # reference, globaldefs.h / mod_grid.F:322+); the wrapper then hands rmask/umask/vmask/pmask to GRID(ng)
  # <APP>_PG31 / <APP>_WJ (UPWELLING, SEAMOUNT): the application's options with DJ_GRADPS taken out / replaced by
  # WJ_GRADP, so that prsgrd.F selects prsgrd31.h (ref_headers/*_pg31.h, *_wj.h)
  # <APP>_DIF4 (UPWELLING, SEAMOUNT): the application's options plus TS_DIF4 and UV_VIS4 (ref_headers/*_dif4.h); the
  # wrapper then also binds t3dmix4 / uv3dmix4 (-DREF_DIF4 reaches the wrapper only)
  local TAG=$1 APP=${1%%_*} XDEF="" WDEF="" VAR=nodiag
  case $TAG in *_MASK*) XDEF="-DMASKING";; esac
  # <APP>[_MASK]_RAD2D: the application with the RADIATION_2D option added on the command line (the tangential phase
  # speed in the radiation conditions of zetabc.F, u2dbc_im.F ... t3dbc_im.F)
  case $TAG in *_RAD2D) XDEF="$XDEF -DRADIATION_2D";; esac
  # <APP>_MASK_WET[...]: + WET_DRY (which needs MASKING).  wetdry.F itself reaches mod_sources -> mod_netcdf and is not
  # part of the build; its callers' WET_DRY blocks (prsgrd*.h, t3dmix*.h, uv3dmix*.h, set_depth.F, mpdata_adiff.F,
  # bulk_flux.F, zetabc.F, u2dbc_im.F ... v3dbc_im.F, ini_fields.F) are, with the masks handed over by the wrapper
  case $TAG in *_WET*) XDEF="$XDEF -DWET_DRY";; esac
  # <APP>_ATM[...]: + ATM_PRESS (the air-pressure term of prsgrd32.h / prsgrd31.h / prsgrd40.h)
  case $TAG in *_ATM*) XDEF="$XDEF -DATM_PRESS";; esac
  case $TAG in *_ATM_PC*) XDEF="$XDEF -DPRESS_COMPENSATE";; esac   # ... and the same term in the Flather value (u2dbc_im.F:264)
  # <APP>_STAB_DIF4 / _STAB_ISO: + TS_MIX_STABILITY (3/4 t(nrhs) + 1/4 t(nstp) in t3dmix2_*.h and the first operator of t3dmix4_*.h)
  case $TAG in *_STAB*) XDEF="$XDEF -DTS_MIX_STABILITY";; esac
  # <APP>_MINSTRAT_ISO: + TS_MIX_MIN_STRAT (the slope scale of t3dmix2_iso.h / t3dmix4_iso.h bounded by a minimum stratification)
  case $TAG in *_MINSTRAT*) XDEF="$XDEF -DTS_MIX_MIN_STRAT";; esac
  case $TAG in *_LIMBS) XDEF="$XDEF -DLIMIT_BSTRESS";; esac
  case $TAG in *_EMP) XDEF="$XDEF -DEMINUSP";; esac              # BENCHMARK[_MASK]_EMP: + EMINUSP (bulk_flux.F:883-899)      # <APP>_LIMBS: + LIMIT_BSTRESS (set_vbc.F:533-567)
  case $TAG in *_PG31) VAR=pg31;; *_WJ) VAR=wj;; *_PJ) VAR=pj;; *_DIF4) VAR=dif4; WDEF="-DREF_DIF4";; *_ISO) VAR=iso; WDEF="-DREF_DIF4";; *_LOGDRAG) VAR=logdrag; WDEF="-DREF_LOGDRAG";; *_GLS) VAR=gls; WDEF="-DREF_GLS";; *_MY25) VAR=my25; WDEF="-DREF_GLS -DREF_MY25";; *_GEOUV) VAR=geouv; WDEF="-DREF_GEOUV";; esac
  local hdr=$(echo $APP | tr A-Z a-z).h
  # UPWELLING: same numerics, output-side options off (see ref_headers/upwelling_nodiag.h)
  [ "$APP" = UPWELLING ] && hdr=upwelling_$VAR.h
  # SEAMOUNT: same numerics without ANA_DIAG, whose ana_diag.h does not compile (see ref_headers/seamount_nodiag.h)
  [ "$APP" = SEAMOUNT ] && hdr=seamount_$VAR.h
  # BENCHMARK_GLS: benchmark.h with the KPP block replaced by GLS_MIXING + CANUTO_A (ref_headers/benchmark_gls.h)
  [ "$APP" = BENCHMARK ] && [ "$VAR" = gls ] && hdr=benchmark_gls.h
  # <APP>[_MASK]_MY25: MY25_MIXING instead (ref_headers/upwelling_my25.h: + KANTHA_CLAYSON + N2S2_HORAVG + RI_SPLINES;
  # benchmark_my25.h: the plain closure with the Galperin functions): my25_prestep.F, my25_corstep.F, tkebc_im.F
  [ "$APP" = BENCHMARK ] && [ "$VAR" = my25 ] && hdr=benchmark_my25.h
  local D=$OUT/$TAG
  if [ -f $D/libref.so ] && [ $D/libref.so -nt $HERE/ref_wrap.F90 ] && [ $D/libref.so -nt $HERE/build_ref.sh ]; then
    return 0
  fi
  mkdir -p $D && cd $D
  local CPPF=(-P -traditional -w -D$APP $XDEF "-DROMS_HEADER=\"$hdr\"" "-DROOT_DIR=\"$REF\"" '-DHOST_NAME="x"'
     '-DMY_OS="Linux"' '-DMY_CPU="x86_64"' '-DMY_FORT="flang"' '-DMY_FC="flang"' '-DMY_FFLAGS="-O2"'
     '-DSVN_URL="x"' '-DSVN_REV="x"' '-DANALYTICAL_DIR="x"' '-DHEADER_DIR="x"' "-DHEADER=\"$hdr\""
     '-DMY_ANALYTICAL_DIR="x"' '-DMY_HEADER_DIR="x"' "-DMY_HEADER=\"$hdr\"" '-DMY_ROOT_DIR="x"' '-DMY_ANALYTICAL="x"'
     -I$HERE/ref_headers -I$REF/ROMS/Include -I$REF/ROMS/Functionals -I$REF/ROMS/Nonlinear -I$REF/ROMS/Utility -I$REF/ROMS/Modules)
  local objs=""
  for f in $FILES; do
    local bn=$(basename $f)
    # the MASKING variants leave analytical.F out: its ana_mask.h stops the compilation on purpose until a user
    # fills in mask values ("no values provided for mask"); the masks reach GRID(ng) through the wrapper
    case "$XDEF" in *MASKING*) [ $bn = analytical ] && continue;; esac
    cpp "${CPPF[@]}" $REF/ROMS/$f.F > $bn.f90
    $FC $FFLAGS -c $bn.f90 -o $bn.o > $bn.log 2>&1 || { echo "[$TAG] $f failed"; tail -5 $bn.log; exit 1; }
    objs="$objs $bn.o"
  done
  cpp -P -traditional -w -D$APP $XDEF $WDEF $HERE/ref_wrap.F90 > ref_wrap_pp.f90
  $FC $FFLAGS -c ref_wrap_pp.f90 -o ref_wrap.o > ref_wrap.log 2>&1 || { echo "[$TAG] ref_wrap failed"; tail -20 ref_wrap.log; exit 1; }
  $FC -shared -o libref.so $objs ref_wrap.o
  rm -f *.f90            # keep no preprocessed reference text around
  echo "[$TAG] built $D/libref.so"
}

for app in ${APPS:-BENCHMARK UPWELLING SEAMOUNT BENCHMARK_MASK UPWELLING_MASK UPWELLING_PG31 UPWELLING_WJ SEAMOUNT_PG31 SEAMOUNT_WJ UPWELLING_DIF4 SEAMOUNT_DIF4 UPWELLING_MASK_DIF4 UPWELLING_ISO SEAMOUNT_ISO UPWELLING_MASK_ISO UPWELLING_LOGDRAG UPWELLING_PJ SEAMOUNT_PJ UPWELLING_RAD2D UPWELLING_MASK_RAD2D BENCHMARK_RAD2D UPWELLING_LIMBS BENCHMARK_LIMBS BENCHMARK_EMP BENCHMARK_MASK_EMP UPWELLING_GLS UPWELLING_MASK_GLS BENCHMARK_GLS UPWELLING_MY25 UPWELLING_MASK_MY25 BENCHMARK_MY25 UPWELLING_GEOUV SEAMOUNT_GEOUV UPWELLING_MASK_GEOUV UPWELLING_MASK_WET_GEOUV UPWELLING_MASK_WET BENCHMARK_MASK_WET UPWELLING_MASK_WET_DIF4 UPWELLING_MASK_WET_ISO UPWELLING_MASK_WET_PG31 UPWELLING_ATM UPWELLING_ATM_PG31 UPWELLING_ATM_PJ UPWELLING_ATM_PC UPWELLING_STAB_DIF4 SEAMOUNT_STAB_DIF4 UPWELLING_STAB_ISO SEAMOUNT_STAB_ISO UPWELLING_MINSTRAT_ISO SEAMOUNT_MINSTRAT_ISO}; do
  build_app $app &
done
wait
