package main

import "math/big"

func newInt(v int64) *big.Int { return big.NewInt(v) }

func modMul(a, b, m *big.Int) *big.Int { r := new(big.Int).Mul(a, b); return r.Mod(r, m) }

func modAdd(a, b, m *big.Int) *big.Int { r := new(big.Int).Add(a, b); return r.Mod(r, m) }
