#!/bin/sh
# One command from a machine with Go (>= 1.13) or Docker to tests/golden/reference_vectors.json -- the file that turns this
# repository's "parity unpinned" into a pinned oracle (tests/test_reference_vectors.py).
#
#   shim/tools/genvectors/run.sh                      # clones github.com/yahoo/bftkv
#   BFTKV_SRC=/path/to/yahoo/bftkv shim/tools/genvectors/run.sh
#   docker build -t genvectors -f shim/tools/genvectors/Dockerfile . && docker run --rm -v "$PWD":/repo genvectors
#
# It builds the reference's OWN crypto/pgp and quorum/wotqs against golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876
# (go.mod here = the reference's go.mod:8 pin; go.sum = the reference's go.sum), runs them over
# tests/golden/reference_inputs.json and writes tests/golden/reference_vectors.json.  Then: python -m pytest tests/test_reference_vectors.py
set -eu
here=$(cd "$(dirname "$0")" && pwd)
repo=$(cd "$here/../../.." && pwd)
work=${GENVECTORS_WORK:-$(mktemp -d)}
mkdir -p "$work"
cp "$here/main.go" "$here/go.mod" "$here/go.sum" "$work/"
if [ -n "${BFTKV_SRC:-}" ]; then
  cp -r "$BFTKV_SRC" "$work/bftkv"
else
  git clone --depth 1 https://github.com/yahoo/bftkv "$work/bftkv"
fi
rm -rf "$work/bftkv/.git"
# the tree every file:line citation of this repository refers to (hashes only; a newer upstream that changed these files is
# reported, not refused: the vectors then pin THAT tree, say so when you commit them)
( cd "$work/bftkv" && sha256sum -c "$here/reference_tree.sha256" ) || echo "genvectors: WARNING: the checkout differs from the tree this repository was written against" >&2
grep -q 'golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876' "$work/bftkv/go.mod" || { echo "genvectors: the checkout does not pin x/crypto at 53104e6ec876" >&2; exit 1; }
patch -p1 --batch -d "$work/bftkv" < "$repo/shim/patches/0001-wotqs-export-cliques.patch"
cd "$work"
# Go >= 1.16 refuses to complete go.sum on its own (the reference is Go 1.13, where that was the default)
case "$(go env GOVERSION 2>/dev/null || true)" in go1.1[6-9]*|go1.[2-9][0-9]*) export GOFLAGS=-mod=mod ;; esac
go run . -in "$repo/tests/golden/reference_inputs.json" -out "$repo/tests/golden/reference_vectors.json"
echo "genvectors: wrote $repo/tests/golden/reference_vectors.json (go: $(go version))"
