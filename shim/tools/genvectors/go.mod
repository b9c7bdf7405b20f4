// genvectors as a module of its own: the reference is `replace`d by a local checkout (run.sh puts one at ./bftkv with
// shim/patches/0001 applied), and golang.org/x/crypto is required at EXACTLY the pseudo-version the reference's go.mod:8 pins.
// go.sum is the reference's own go.sum, byte for byte (its hashes cover every module of this build list).
module github.com/bftkv-amd/genvectors

go 1.13

require (
	github.com/yahoo/bftkv v0.0.0-00010101000000-000000000000
	golang.org/x/crypto v0.0.0-20191227163750-53104e6ec876
)

replace github.com/yahoo/bftkv => ./bftkv
