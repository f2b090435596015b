module github.com/giuliop/algoplonk/tools/gnark_dump

go 1.23

// the versions /root/reference/go.mod:8-9 pins
require (
	github.com/consensys/gnark v0.15.0
	github.com/consensys/gnark-crypto v0.20.1
)
