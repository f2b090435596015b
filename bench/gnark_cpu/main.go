// gnark_cpu - the REAL reference CPU path for bench.py's cpu_baseline (BASELINE.md B1, SURVEY.md section 8d plan (1)).
//
// Times gnark v0.15.0 plonk.Prove - the call AlgoPlonk makes at algoplonk.go:89 - on a circuit of 2^log-n constraints with
// an unsafekzg SRS (what setup.Run does for the TestOnly setups, setup/setup.go:102-108), GOMAXPROCS = all host cores,
// and prints ONE JSON line: {"proofs_per_sec", "proofs", "seconds", "cores", "curve", "log_n", "nb_constraints", "go", "cpu"}.
//
// bench.py probes for a Go toolchain (`go version`) and a module cache holding the two modules (`go build` with
// GOFLAGS=-mod=mod GOPROXY=off); when both are there it builds this program into oracle/_ref/gnark_cpu and reports its number
// as cpu_baseline.kind = "reference".  NOT COMPILED in the build container (no Go toolchain there): source-only
// deliverable, written against the gnark v0.15.0 API the reference itself uses (algoplonk.go:50,81,89; setup/setup.go:103-107).
package main

import (
	"encoding/json"
	"flag"
	"fmt"
	"os"
	"runtime"
	"time"

	"github.com/consensys/gnark-crypto/ecc"
	"github.com/consensys/gnark/backend/plonk"
	"github.com/consensys/gnark/constraint"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/scs"
	"github.com/consensys/gnark/test/unsafekzg"
)

// chain: x_{i+1} = x_i * x_i + x_i * Y + i  (one mul-add gate per step, like bench.py's random gates
// c = ql*a + qr*b + qm*a*b + qk); the final value is public so the circuit has two public inputs as configs[1] does.
type chain struct {
	X     frontend.Variable
	Y     frontend.Variable `gnark:",public"`
	Out   frontend.Variable `gnark:",public"`
	steps int
}

func (c *chain) Define(api frontend.API) error {
	x := c.X
	for i := 0; i < c.steps; i++ {
		// x*x + x*Y + i : scs folds this into two constraints (mul, then mul-add)
		x = api.Add(api.Mul(x, x), api.Mul(x, c.Y), i)
	}
	api.AssertIsEqual(x, c.Out)
	return nil
}

func main() {
	curveName := flag.String("curve", "bn254", "bn254 | bls12_381")
	logN := flag.Int("log-n", 17, "log2 of the PLONK domain size to fill")
	seconds := flag.Float64("seconds", 20, "time budget for the timed proofs")
	flag.Parse()

	curve := ecc.BN254
	if *curveName == "bls12_381" {
		curve = ecc.BLS12_381
	}
	field := curve.ScalarField()

	// size the chain so that nbConstraints + nbPublic lands just under 2^log-n (setup/setup.go:113-114 rounds up)
	target := (1 << *logN) - 8
	steps := target / 2
	var ccsSize int
	var circuit chain
	for {
		circuit = chain{steps: steps}
		ccs, err := frontend.Compile(field, scs.NewBuilder, &circuit)
		if err != nil {
			fail(err)
		}
		ccsSize = ccs.GetNbConstraints() + ccs.GetNbPublicVariables()
		if ccsSize <= 1<<*logN && ccsSize > 1<<(*logN-1) {
			run(ccs, steps, curve, *logN, ccsSize, *seconds)
			return
		}
		// adjust the step count proportionally and retry (the builder's folding decides the exact ratio)
		steps = steps * target / ccsSize
		if steps < 1 {
			fail(fmt.Errorf("cannot size the circuit"))
		}
	}
}

func run(cs constraint.ConstraintSystem, steps int, curve ecc.ID, logN, size int, seconds float64) {
	field := curve.ScalarField()
	srs, lagrange, err := unsafekzg.NewSRS(cs) // setup/setup.go:103
	if err != nil {
		fail(err)
	}
	pk, vk, err := plonk.Setup(cs, srs, lagrange) // setup/setup.go:107
	if err != nil {
		fail(err)
	}
	// witness: X = 3, Y = 5, Out = the chain's value (computed in the field with big.Int arithmetic)
	assignment := chain{steps: steps}
	x, y := newInt(3), newInt(5)
	for i := 0; i < steps; i++ {
		x = modAdd(modAdd(modMul(x, x, field), modMul(x, y, field), field), newInt(int64(i)), field)
	}
	assignment.X, assignment.Y, assignment.Out = newInt(3), y, x
	w, err := frontend.NewWitness(&assignment, field) // algoplonk.go:81
	if err != nil {
		fail(err)
	}
	pub, _ := w.Public()
	// warm-up (also checks the proof verifies: algoplonk.go:93)
	proof, err := plonk.Prove(cs, pk, w) // algoplonk.go:89
	if err != nil {
		fail(err)
	}
	if err := plonk.Verify(proof, vk, pub); err != nil {
		fail(err)
	}
	done := 0
	t0 := time.Now()
	for {
		if _, err := plonk.Prove(cs, pk, w); err != nil {
			fail(err)
		}
		done++
		el := time.Since(t0).Seconds()
		if el >= seconds || el+el/float64(done) > 1.5*seconds {
			break
		}
	}
	el := time.Since(t0).Seconds()
	out := map[string]interface{}{
		"proofs_per_sec": float64(done) / el, "proofs": done, "seconds": el, "cores": runtime.GOMAXPROCS(0),
		"curve": curve.String(), "log_n": logN, "nb_constraints": size, "go": runtime.Version(), "prover": "gnark v0.15.0 plonk.Prove",
	}
	b, _ := json.Marshal(out)
	fmt.Println(string(b))
}

func fail(err error) {
	fmt.Fprintln(os.Stderr, "gnark_cpu:", err)
	os.Exit(1)
}
