R=$GRAFT_REPO_ROOT; O=$R/gpurun_out/r04_pk4; mkdir -p $O; cd $R
for defs in "-DPK_CDMA=0 -DPK_PRIO=0" "-DPK_CDMA=0 -DPK_PRIO=1"; do
  echo "#### $defs"
  PK_DEFS="$defs" bash tools/r04_pkstamps.sh 128 > $O/st.txt 2>&1; head -14 $O/st.txt; grep -A 18 "arrival at the stage barrier" $O/st.txt
done
bash tools/r04_pkab.sh "-DPK_CDMA=0 -DPK_PRIO=0" "-DPK_CDMA=0 -DPK_PRIO=1" "-DPK_CDMA=0 -DPK_PRIO=0 -DPK_INTERIOR=0"
