for lib in prisma_amd/libprisma_bands_base.so prisma_amd/libprisma_bands.so; do
  PRISMA_BANDS_LIB=$PWD/$lib AB_PREC=1 python tools/ab_flow.py 2>&1 | tail -1 | python -c "
import sys,json,re
l=sys.stdin.read(); j=json.loads(l[l.index('{'):]); print('$lib', 'total', l.split('total')[1].split('ms')[0], 'halo', j.get('conv3x3_c64_mx2_kernel'))"
done
