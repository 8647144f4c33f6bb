// Std-only Rust restatement of the reference's CPU scan for bench.py's run-time `rustc` probe: the inner
// product of src/search/vector.rs:99-101 (sequential f32 sum), distance = 1 - dot (:128-134), and the keep-10-best loop of
// examples_old/search.rs:49-72 over a packed [N][384] f32 array.  Synthetic unit rows are generated in place (xorshift;
// the data only has to be unit-length for the timing to be representative).  Prints rows/s on one thread.
use std::time::Instant;

const EM: usize = 384;

fn distance_ip(a: &[f32], b: &[f32]) -> f32 {
    let mut result = 0.0f32;
    for i in 0..EM {
        result += a[i] * b[i];
    }
    result
}

fn main() {
    let n: usize = std::env::args().nth(1).and_then(|s| s.parse().ok()).unwrap_or(1_000_000);
    let mut state: u64 = 0x9E3779B97F4A7C15;
    let mut next = || {
        state ^= state << 13;
        state ^= state >> 7;
        state ^= state << 17;
        ((state >> 40) as f32) / 8388608.0 - 1.0
    };
    let mut rows = vec![0f32; n * EM];
    for r in 0..n {
        let row = &mut rows[r * EM..(r + 1) * EM];
        let mut s = 0.0f32;
        for v in row.iter_mut() {
            *v = next();
            s += *v * *v;
        }
        let l = s.sqrt();
        for v in row.iter_mut() {
            *v /= l;
        }
    }
    let q: Vec<f32> = rows[7 * EM..8 * EM].to_vec();
    let t0 = Instant::now();
    let mut results: Vec<(f32, usize)> = Vec::with_capacity(11);
    for r in 0..n {
        let score = 1.0 - distance_ip(&rows[r * EM..(r + 1) * EM], &q);
        if results.len() < 10 {
            results.push((score, r));
            continue;
        }
        if score < results[9].0 {
            results[9] = (score, r);
            results.sort_by(|a, b| a.0.partial_cmp(&b.0).unwrap());
        }
    }
    let dt = t0.elapsed().as_secs_f64();
    println!("{{\"rows\": {}, \"seconds\": {:.6}, \"rows_per_s\": {:.1}, \"top1\": {}}}", n, dt, n as f64 / dt, results[0].1);
}
