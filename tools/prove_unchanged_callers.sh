#!/bin/bash
# One-off evidence run (build container -> GPU box): the UNMODIFIED reference Workflow.py / GraphGenerator.py driving the
# MI355X drop-ins on a real MI355X.  The reference checkout is not on the GPU box and its sources are not part of this
# repository, so the two files are STAGED next to the snapshot for this one gpurun call (.refstage/ is git-ignored) and
# removed again whatever happens; tests/test_callers_gpu.py then takes its "unmodified reference callers" branch
# (GI_REFERENCE_DIR).  The log (with the sha256 of the staged files = those of /root/reference) goes to
# profiles/<tag>/unchanged_callers_gpu.log.
TAG=${1:-r04}
cd /root/repo
STAGE=.refstage/graphinvent
trap 'rm -rf /root/repo/.refstage' EXIT
mkdir -p $STAGE gpurun_out/$TAG
cp /root/reference/graphinvent/Workflow.py /root/reference/graphinvent/GraphGenerator.py $STAGE/
( cd /root/reference/graphinvent && sha256sum Workflow.py GraphGenerator.py ) > gpurun_out/$TAG/ref_sha256.txt
/usr/local/graft/bin/gpurun --timeout 600 -- "mkdir -p gpurun_out/$TAG; ( echo '# sha256 of the staged files on the GPU box:'; cd .refstage/graphinvent && sha256sum Workflow.py GraphGenerator.py; cd ../..; GI_REFERENCE_DIR=\$PWD/.refstage/graphinvent timeout 500 python -m pytest tests/test_callers_gpu.py -q -m gpu -s 2>&1 | grep -v 'it/s\]' ) > gpurun_out/$TAG/unchanged_callers_gpu.log 2>&1; tail -25 gpurun_out/$TAG/unchanged_callers_gpu.log"
echo "# sha256 of the reference's files in the build container:"; cat gpurun_out/$TAG/ref_sha256.txt
