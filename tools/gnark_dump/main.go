// gnark_dump - the ONE program that can pin this repo's prover against gnark at the value level (SURVEY.md section 8c (iv),
// App. D; VERDICT r01 item 1d).  It runs gnark v0.15.0's BN254 prover - the call AlgoPlonk makes at algoplonk.go:89 - with
// crypto/rand replaced by a seeded stream, and dumps EVERYTHING that crosses libapk's C-ABI plus what gnark produced:
//
//   inputs  : tau, the canonical SRS (n+3 G1) and Lagrange SRS (n G1) in gnark's in-memory bytes; the trace gnark built
//             (Ql,Qr,Qm,Qo,Qk in Lagrange form, S) = apk_circuit_desc; the solved L,R,O and the public witness = apk_prove's
//             arguments; the field elements gnark's fr.Element.SetRandom() yields from the seeded stream, in draw order
//             (the candidates for apk_prove's 9 blinding scalars: App. D.1 asks in which order gnark consumes them)
//   outputs : every field of gnark's Proof (LRO, Z, H, BatchedProof, ZShiftedOpening) in in-memory bytes, MarshalSolidity
//             (= helper.go:17, the 768-byte AVM blob), the VK commitments, and two primitives on their own: one
//             kzg.Commit (MSM) and one fft.Domain.FFT / FFTInverse (NTT) of a seeded vector.
//
// Output: one JSON file in the format tests/golden/ uses (hex strings), consumed by tests/test_gnark_dump.py: the test feeds
// the dumped inputs to libapk and compares bytes.  Usage:  go run . -log-n 6 -seed 0xA190 -o ../../tests/golden/gnark_bn254_2p6.json
//
// NOT COMPILED in the build container (no Go toolchain, no module cache): source-only deliverable.  It uses gnark internals
// that are exported but unstable (plonk_bn254.NewTrace, cs.SparseR1CS.Solve) - the same ones INTEGRATION.md's shim binds.
package main

import (
	"bytes"
	"crypto/rand"
	"crypto/sha256"
	"encoding/binary"
	"encoding/hex"
	"encoding/json"
	"flag"
	"fmt"
	"math/big"
	"os"
	"unsafe"

	"github.com/consensys/gnark-crypto/ecc"
	curve "github.com/consensys/gnark-crypto/ecc/bn254"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr"
	"github.com/consensys/gnark-crypto/ecc/bn254/fr/fft"
	"github.com/consensys/gnark-crypto/ecc/bn254/kzg"
	"github.com/consensys/gnark/backend/plonk"
	plonk_bn254 "github.com/consensys/gnark/backend/plonk/bn254"
	cs_bn254 "github.com/consensys/gnark/constraint/bn254"
	"github.com/consensys/gnark/frontend"
	"github.com/consensys/gnark/frontend/cs/scs"
)

// seeded byte stream: SHA-256 in counter mode.  Installed as crypto/rand.Reader, so every fr.Element.SetRandom() inside
// gnark's prover (the blinding polynomials; the BSB22 hiding entries) reads from it.
type stream struct {
	seed [32]byte
	ctr  uint64
	buf  []byte
}

func newStream(seed uint64) *stream {
	s := &stream{}
	binary.BigEndian.PutUint64(s.seed[24:], seed)
	return s
}

func (s *stream) Read(p []byte) (int, error) {
	for i := range p {
		if len(s.buf) == 0 {
			var c [8]byte
			binary.BigEndian.PutUint64(c[:], s.ctr)
			s.ctr++
			h := sha256.Sum256(append(s.seed[:], c[:]...))
			s.buf = h[:]
		}
		p[i] = s.buf[0]
		s.buf = s.buf[1:]
	}
	return len(p), nil
}

// x*x + x*Y + i chain with two public inputs (same shape as bench/gnark_cpu)
type chain struct {
	X     frontend.Variable
	Y     frontend.Variable `gnark:",public"`
	Out   frontend.Variable `gnark:",public"`
	steps int
}

func (c *chain) Define(api frontend.API) error {
	x := c.X
	for i := 0; i < c.steps; i++ {
		x = api.Add(api.Mul(x, x), api.Mul(x, c.Y), i)
	}
	api.AssertIsEqual(x, c.Out)
	return nil
}

func raw(p unsafe.Pointer, n int) string { return hex.EncodeToString(unsafe.Slice((*byte)(p), n)) }
func frs(v []fr.Element) string           { return raw(unsafe.Pointer(&v[0]), 32*len(v)) }
func g1s(v []curve.G1Affine) string       { return raw(unsafe.Pointer(&v[0]), 64*len(v)) }
func g1(p *curve.G1Affine) string         { return raw(unsafe.Pointer(p), 64) }

func main() {
	logN := flag.Int("log-n", 6, "log2 of the domain size")
	seed := flag.Uint64("seed", 0xA190, "seed of the blinding stream and of tau")
	outPath := flag.String("o", "gnark_bn254.json", "output file")
	flag.Parse()
	field := ecc.BN254.ScalarField()
	n := uint64(1) << *logN

	// circuit sized to land in (n/2, n]
	steps := int(n)/2 - 4
	var ccs *cs_bn254.SparseR1CS
	for {
		c := chain{steps: steps}
		cc, err := frontend.Compile(field, scs.NewBuilder, &c)
		must(err)
		size := uint64(cc.GetNbConstraints() + cc.GetNbPublicVariables())
		if size <= n && size > n/2 {
			ccs = cc.(*cs_bn254.SparseR1CS)
			break
		}
		if size > n {
			steps--
		} else {
			steps++
		}
	}

	// SRS from a known tau = SHA-256(seed) mod r (what algoplonk_amd/workloads.py::tau_from_seed computes), built the way
	// setup.Run builds it: canonical n+3 points, Lagrange from the first n (setup/setup.go:113-143)
	var sb [8]byte
	binary.BigEndian.PutUint64(sb[:], *seed)
	th := sha256.Sum256(sb[:])
	tau := new(big.Int).Mod(new(big.Int).SetBytes(th[:]), field)
	srs, err := kzg.NewSRS(n+3, tau)
	must(err)
	lag := &kzg.SRS{Vk: srs.Vk}
	lag.Pk.G1, err = kzg.ToLagrangeG1(srs.Pk.G1[:n])
	must(err)
	pkI, vkI, err := plonk.Setup(ccs, srs, lag) // setup/setup.go:107,149
	must(err)
	pk, vk := pkI.(*plonk_bn254.ProvingKey), vkI.(*plonk_bn254.VerifyingKey)

	// witness
	x, y := big.NewInt(3), big.NewInt(5)
	for i := 0; i < steps; i++ {
		t := new(big.Int).Mul(x, x)
		t.Add(t, new(big.Int).Mul(x, y)).Add(t, big.NewInt(int64(i)))
		x = t.Mod(t, field)
	}
	w, err := frontend.NewWitness(&chain{X: 3, Y: 5, Out: x, steps: steps}, field)
	must(err)
	pub, err := w.Public()
	must(err)

	// what crosses the C-ABI: the trace and the solved wires
	domain := fft.NewDomain(n)
	trace := plonk_bn254.NewTrace(ccs, domain)
	solI, err := ccs.Solve(w)
	must(err)
	sol := solI.(*cs_bn254.SparseR1CSSolution)
	perm := make([]int64, len(trace.S))
	copy(perm, trace.S)

	// the draws gnark will see: same stream, same seed, consumed through the same fr.Element.SetRandom()
	rand.Reader = newStream(*seed)
	draws := make([]fr.Element, 16)
	for i := range draws {
		_, err := draws[i].SetRandom()
		must(err)
	}
	// the proof, with the stream rewound
	rand.Reader = newStream(*seed)
	proofI, err := plonk.Prove(ccs, pk, w) // algoplonk.go:89
	must(err)
	must(plonk.Verify(proofI, vk, pub)) // algoplonk.go:93
	proof := proofI.(*plonk_bn254.Proof)

	// primitives on their own: kzg.Commit of the L column's canonical form, and one FFT round trip
	vec := make([]fr.Element, n)
	copy(vec, sol.L)
	canon := make([]fr.Element, n)
	copy(canon, vec)
	domain.FFTInverse(canon, fft.DIF)
	fft.BitReverse(canon)
	commit, err := kzg.Commit(canon, srs.Pk)
	must(err)
	evals := make([]fr.Element, n)
	copy(evals, canon)
	domain.FFT(evals, fft.DIF)
	fft.BitReverse(evals) // == vec

	// gnark's own serialisations (what utils.SerializeCompiledCircuit wraps, utils/utils.go:97-122): pins algoplonk_amd/serialize.py
	var vkBuf, ccsBuf bytes.Buffer
	_, err = vk.WriteTo(&vkBuf)
	must(err)
	_, err = ccs.WriteTo(&ccsBuf)
	must(err)

	pubVec := pub.Vector().(fr.Vector)
	claimed := proof.BatchedProof.ClaimedValues
	out := map[string]interface{}{
		"source": "tools/gnark_dump (gnark v0.15.0, gnark-crypto v0.20.1)", "curve": "bn254", "log_n": *logN, "seed": *seed,
		"layout": "hex of gnark's in-memory bytes: fr.Element = 4 little-endian u64 limbs, Montgomery; G1Affine = X || Y",
		"tau": tau.Text(16), "nb_public": ccs.GetNbPublicVariables(), "nb_constraints": ccs.GetNbConstraints(),
		"srs_g1": g1s(srs.Pk.G1), "srs_g1_lagrange": g1s(lag.Pk.G1),
		"ql": frs(trace.Ql.Coefficients()), "qr": frs(trace.Qr.Coefficients()), "qm": frs(trace.Qm.Coefficients()),
		"qo": frs(trace.Qo.Coefficients()), "qk": frs(trace.Qk.Coefficients()), "perm": perm,
		"L": frs(sol.L), "R": frs(sol.R), "O": frs(sol.O), "public": frs(pubVec),
		"set_random_draws": frs(draws),
		"proof": map[string]interface{}{
			"lro": []string{g1(&proof.LRO[0]), g1(&proof.LRO[1]), g1(&proof.LRO[2])}, "z": g1(&proof.Z),
			"h": []string{g1(&proof.H[0]), g1(&proof.H[1]), g1(&proof.H[2])}, "batched_h": g1(&proof.BatchedProof.H),
			"claimed_values": frs(claimed), "zshift_h": g1(&proof.ZShiftedOpening.H),
			"zshift_value": frs([]fr.Element{proof.ZShiftedOpening.ClaimedValue}),
			"marshal_solidity": hex.EncodeToString(proof.MarshalSolidity()),
		},
		"vk": map[string]interface{}{
			"ql": g1(&vk.Ql), "qr": g1(&vk.Qr), "qm": g1(&vk.Qm), "qo": g1(&vk.Qo), "qk": g1(&vk.Qk),
			"s": []string{g1(&vk.S[0]), g1(&vk.S[1]), g1(&vk.S[2])},
			"size_inv": frs([]fr.Element{vk.SizeInv}), "generator": frs([]fr.Element{vk.Generator}), "coset_shift": frs([]fr.Element{vk.CosetShift}),
		},
		"vk_write_to": hex.EncodeToString(vkBuf.Bytes()), "ccs_write_to_len": ccsBuf.Len(), "ecc_id": uint16(ecc.BN254),
		"primitives": map[string]interface{}{
			"msm_scalars": frs(canon), "msm_commit": g1(&commit), "ntt_in": frs(canon), "ntt_out": frs(evals),
		},
	}
	f, err := os.Create(*outPath)
	must(err)
	defer f.Close()
	enc := json.NewEncoder(f)
	enc.SetIndent("", " ")
	must(enc.Encode(out))
	fmt.Println("wrote", *outPath, "n =", n, "constraints =", ccs.GetNbConstraints())
}

func must(err error) {
	if err != nil {
		fmt.Fprintln(os.Stderr, "gnark_dump:", err)
		os.Exit(1)
	}
}
