// GoldenParity.cs -- pins this repo's oracle (and, when libhare_hip.so + a GPU are present, the HIP kernels) to the
// REFERENCE implementation: runs PachydermAcoustic/Hare's own Voxel_Grid / Octree / KDTree .Shoot on the committed
// config-1 inputs (and on case 2: exact ties, trees over two topologies; case 3: quadrilaterals) and compares every X_Event with the committed
// expected records, bit for bit.
//
// It cannot run where this repo is built (no .NET toolchain there; DESIGN.md section 1) -- it is the recipe for
// whoever has `dotnet`:
//
//     python tests/golden/export_raw.py                      # writes tests/golden/raw/*.f64 *.i32 *.xev params.txt
//     cd <a checkout of PachydermAcoustic/Hare>
//     cp <this repo>/bindings/csharp/tests/GoldenParity.cs .      (plus HareHip.cs, Gpu_Spatial_Partition.cs for --gpu)
//     dotnet new console -n GoldenParity -o gp && cp GoldenParity.cs gp/Program.cs && \
//         dotnet add gp reference Hare.csproj && dotnet run --project gp -- <this repo>/tests/golden/raw [--gpu]
//
// Exit code 0 = every record of every partition identical (the oracle is then pinned by the reference itself);
// otherwise the first differing rays are printed per partition and field.
//
// What it does per partition P in {Voxel_Grid(D), Octree(OD, OP), KDTree(KDD, KDP)}:
//     Topology T = new Topology(Point[][] tris); T.Finish_Topology();     (Hare_Geometry_Topology.cs:121, :148)
//     for ray i:  R = new Ray(x, y, z, dx, dy, dz, 0, pass * 1000000 + i + 1)   -- never Ray_ID 0 (SURVEY.md F7)
//                 P.Shoot(R, 0, out X_Event e)                                -- Spatial_Partition.cs:32
//                 P.Shoot(R, 0, out X_Event e, excl1[i])                      -- Spatial_Partition.cs:33 (voxel, octree)
// A reference call that throws (the C# indexes out of range for some rays that leave the grid at the clip point,
// Voxel_Grid.cs:584-593) is compared against the miss record, which is what the oracle and the library return there.
using System;
using System.Collections.Generic;
using System.IO;
using Hare.Geometry;

public static class GoldenParity
{
    struct Rec { public double t, u, v, x, y, z; public int poly, hit; }

    static double[] ReadF64(string p) { byte[] b = File.ReadAllBytes(p); var a = new double[b.Length / 8]; Buffer.BlockCopy(b, 0, a, 0, b.Length); return a; }
    static int[] ReadI32(string p) { byte[] b = File.ReadAllBytes(p); var a = new int[b.Length / 4]; Buffer.BlockCopy(b, 0, a, 0, b.Length); return a; }

    static Rec[] ReadEvents(string p, int n)
    {
        byte[] b = File.ReadAllBytes(p);
        if (b.Length != n * 56) throw new InvalidDataException(p + ": expected " + (n * 56) + " bytes");
        var r = new Rec[n];
        for (int i = 0; i < n; i++)
        {
            int o = i * 56;
            r[i].t = BitConverter.ToDouble(b, o); r[i].u = BitConverter.ToDouble(b, o + 8); r[i].v = BitConverter.ToDouble(b, o + 16);
            r[i].x = BitConverter.ToDouble(b, o + 24); r[i].y = BitConverter.ToDouble(b, o + 32); r[i].z = BitConverter.ToDouble(b, o + 40);
            r[i].poly = BitConverter.ToInt32(b, o + 48); r[i].hit = BitConverter.ToInt32(b, o + 52);
        }
        return r;
    }

    static Rec FromEvent(X_Event e)
    {
        var r = new Rec();
        if (e != null && e.Hit)
        {
            r.t = e.t; r.u = e.u; r.v = e.v; r.x = e.X_Point.x; r.y = e.X_Point.y; r.z = e.X_Point.z; r.poly = e.Poly_id; r.hit = 1;
        }
        else { r.poly = -1; }     // X_Event(): Hare_Geometry_Primitives.cs:454-462
        return r;
    }

    static bool Same(double a, double b) { return BitConverter.DoubleToInt64Bits(a) == BitConverter.DoubleToInt64Bits(b); }

    /// <returns>number of rays whose record differs in any field</returns>
    static int Compare(string what, Rec[] got, Rec[] want, int threw)
    {
        int bad = 0;
        for (int i = 0; i < want.Length; i++)
        {
            Rec g = got[i], w = want[i];
            bool ok = g.hit == w.hit && g.poly == w.poly && Same(g.t, w.t) && Same(g.u, w.u) && Same(g.v, w.v) && Same(g.x, w.x) && Same(g.y, w.y) && Same(g.z, w.z);
            if (!ok && bad++ < 5)
                Console.WriteLine("  {0}: ray {1}: reference (hit {2} poly {3} t {4:R} u {5:R} v {6:R} X {7:R},{8:R},{9:R})  expected (hit {10} poly {11} t {12:R} u {13:R} v {14:R} X {15:R},{16:R},{17:R})",
                                  what, i, g.hit, g.poly, g.t, g.u, g.v, g.x, g.y, g.z, w.hit, w.poly, w.t, w.u, w.v, w.x, w.y, w.z);
        }
        Console.WriteLine("{0,-14} {1} / {2} records identical{3}", what, want.Length - bad, want.Length, threw > 0 ? "  (" + threw + " reference calls threw; compared as misses)" : "");
        return bad;
    }

    static int Run(string what, Spatial_Partition part, double[] rays, int[] excl, Rec[] want, int pass) { return Run(what, part, rays, excl, want, pass, 0); }

    static int Run(string what, Spatial_Partition part, double[] rays, int[] excl, Rec[] want, int pass, int top)
    {
        int n = want.Length, threw = 0;
        var got = new Rec[n];
        for (int i = 0; i < n; i++)
        {
            var R = new Ray(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2], rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5], 0, pass * 1000000 + i + 1);
            X_Event e = null;
            try
            {
                if (excl == null) part.Shoot(R, top, out e);
                else part.Shoot(R, top, out e, excl[i]);
            }
            catch (Exception) { e = null; threw++; }
            got[i] = FromEvent(e);
        }
        return Compare(what, got, want, threw);
    }

    static Topology MakeTopology(double[] tris, int P)
    {
        var polys = new Point[P][];
        for (int k = 0; k < P; k++)
        {
            polys[k] = new Point[3];
            for (int c = 0; c < 3; c++) polys[k][c] = new Point(tris[9 * k + 3 * c], tris[9 * k + 3 * c + 1], tris[9 * k + 3 * c + 2]);
        }
        var T = new Topology(polys);
        T.Finish_Topology();
        if (T.Polygon_Count != P) throw new InvalidDataException("Topology kept " + T.Polygon_Count + " of " + P + " polygons");
        return T;
    }

    // Case 2 (tests/golden/c2_ties.npz, written to <dir>/c2 by export_raw.py): rays aimed exactly at corners and edges, coincident
    // and coplanar overlapping triangles (strict `<`, list order), and Octree / KDTree built over TWO topologies, shot at
    // top_index 0 and 1 ("Octree - alt.cs":63-88,123; KDTree.cs:67-87).
    static int Case2(string dir)
    {
        if (!File.Exists(Path.Combine(dir, "params.txt"))) { Console.WriteLine("case 2 not exported (" + dir + "): skipped"); return 0; }
        string[] p = File.ReadAllText(Path.Combine(dir, "params.txt")).Split(new[] { ' ', '\n', '\r' }, StringSplitOptions.RemoveEmptyEntries);
        int D = int.Parse(p[0]), OD = int.Parse(p[1]), OP = int.Parse(p[2]), KDD = int.Parse(p[3]), KDP = int.Parse(p[4]);
        int P0 = int.Parse(p[5]), P1 = int.Parse(p[6]), N = int.Parse(p[7]);
        double[] rays = ReadF64(Path.Combine(dir, "rays.f64"));
        int[] excl1 = ReadI32(Path.Combine(dir, "excl1.i32")), excl2 = ReadI32(Path.Combine(dir, "excl1_two.i32"));
        Topology T0 = MakeTopology(ReadF64(Path.Combine(dir, "tris0.f64")), P0), T1 = MakeTopology(ReadF64(Path.Combine(dir, "tris1.f64")), P1);
        var one = new Topology[] { T0 };
        var two = new Topology[] { T0, T1 };
        Func<string, Rec[]> ev = name => ReadEvents(Path.Combine(dir, name + ".xev"), N);
        int bad = 0;
        var vox = new Voxel_Grid(one, D);
        bad += Run("c2 voxel", vox, rays, null, ev("voxel"), 11);
        bad += Run("c2 voxel_excl", vox, rays, excl1, ev("voxel_excl"), 12);
        var oct = new Octree(one, OD, OP);
        bad += Run("c2 octree", oct, rays, null, ev("octree"), 13);
        bad += Run("c2 octree_excl", oct, rays, excl1, ev("octree_excl"), 14);
        bad += Run("c2 kdtree", new KDTree(one, KDD, KDP), rays, null, ev("kdtree"), 15);
        var oct2 = new Octree(two, OD, OP);
        var kd2 = new KDTree(two, KDD, KDP);
        for (int top = 0; top < 2; top++)
        {
            bad += Run("c2 octree2 top" + top, oct2, rays, null, ev("octree2_top" + top), 16 + 3 * top, top);
            bad += Run("c2 octree2x top" + top, oct2, rays, excl2, ev("octree2_top" + top + "_excl"), 17 + 3 * top, top);
            bad += Run("c2 kdtree2 top" + top, kd2, rays, null, ev("kdtree2_top" + top), 18 + 3 * top, top);
        }
        return bad;
    }

    // Case 3 (tests/golden/c3_quads.npz, written to <dir>/c3 by export_raw.py): QUADRILATERALS -- a room of rectangles, general convex
    // and tilted quadrilaterals, coincident twins, mixed with triangles; rays aimed at corners, edge points, the diagonal both triangles of
    // Quadrilateral.Intersect share (Hare_Geometry_Polygons.cs:731-823), and interiors.  Topology(Point[][]) makes a Quadrilateral of a
    // four-corner polygon (Hare_Geometry_Topology.cs:288-291).
    static int Case3(string dir)
    {
        if (!File.Exists(Path.Combine(dir, "params.txt"))) { Console.WriteLine("case 3 not exported (" + dir + "): skipped"); return 0; }
        string[] p = File.ReadAllText(Path.Combine(dir, "params.txt")).Split(new[] { ' ', '\n', '\r' }, StringSplitOptions.RemoveEmptyEntries);
        int D = int.Parse(p[0]), OD = int.Parse(p[1]), OP = int.Parse(p[2]), KDD = int.Parse(p[3]), KDP = int.Parse(p[4]), P = int.Parse(p[5]), N = int.Parse(p[6]);
        double[] v = ReadF64(Path.Combine(dir, "polys.f64"));
        int[] nv = ReadI32(Path.Combine(dir, "nverts.i32"));
        double[] rays = ReadF64(Path.Combine(dir, "rays.f64"));
        int[] excl1 = ReadI32(Path.Combine(dir, "excl1.i32"));
        if (v.Length != P * 12 || nv.Length != P || rays.Length != N * 6 || excl1.Length != N) throw new InvalidDataException("case 3: sizes disagree with params.txt");
        var polys = new Point[P][];
        for (int k = 0; k < P; k++)
        {
            polys[k] = new Point[nv[k]];
            for (int c = 0; c < nv[k]; c++) polys[k][c] = new Point(v[12 * k + 3 * c], v[12 * k + 3 * c + 1], v[12 * k + 3 * c + 2]);
        }
        var T = new Topology(polys);
        T.Finish_Topology();
        if (T.Polygon_Count != P) throw new InvalidDataException("case 3: Topology kept " + T.Polygon_Count + " of " + P + " polygons");
        var model = new Topology[] { T };
        Func<string, Rec[]> ev = name => ReadEvents(Path.Combine(dir, name + ".xev"), N);
        int bad = 0;
        var vox = new Voxel_Grid(model, D);
        bad += Run("c3 voxel", vox, rays, null, ev("voxel"), 31);
        bad += Run("c3 voxel_excl", vox, rays, excl1, ev("voxel_excl"), 32);
        var oct = new Octree(model, OD, OP);
        bad += Run("c3 octree", oct, rays, null, ev("octree"), 33);
        bad += Run("c3 octree_excl", oct, rays, excl1, ev("octree_excl"), 34);
        bad += Run("c3 kdtree", new KDTree(model, KDD, KDP), rays, null, ev("kdtree"), 35);
        return bad;
    }

    public static int Main(string[] args)
    {
        if (args.Length < 1) { Console.Error.WriteLine("usage: GoldenParity <dir written by tests/golden/export_raw.py> [--gpu]"); return 2; }
        string dir = args[0];
        bool gpu = args.Length > 1 && args[1] == "--gpu";
        string[] p = File.ReadAllText(Path.Combine(dir, "params.txt")).Split(new[] { ' ', '\n', '\r' }, StringSplitOptions.RemoveEmptyEntries);
        int D = int.Parse(p[0]), OD = int.Parse(p[1]), OP = int.Parse(p[2]), KDD = int.Parse(p[3]), KDP = int.Parse(p[4]), P = int.Parse(p[5]), N = int.Parse(p[6]);
        double[] tris = ReadF64(Path.Combine(dir, "tris.f64"));
        double[] rays = ReadF64(Path.Combine(dir, "rays.f64"));
        int[] excl1 = ReadI32(Path.Combine(dir, "excl1.i32"));
        if (tris.Length != P * 9 || rays.Length != N * 6 || excl1.Length != N) throw new InvalidDataException("sizes disagree with params.txt");

        var polys = new Point[P][];
        for (int k = 0; k < P; k++)
        {
            polys[k] = new Point[3];
            for (int c = 0; c < 3; c++) polys[k][c] = new Point(tris[9 * k + 3 * c], tris[9 * k + 3 * c + 1], tris[9 * k + 3 * c + 2]);
        }
        // every coordinate is on the 2^-8 m lattice, so the ingest (Math.Round(x, 15) + 1 mm Hash2 merge) leaves it bit-identical
        var T = new Topology(polys);
        T.Finish_Topology();
        if (T.Polygon_Count != P) throw new InvalidDataException("Topology kept " + T.Polygon_Count + " of " + P + " polygons");
        var model = new Topology[] { T };

        int bad = 0;
        var vox = new Voxel_Grid(model, D);
        bad += Run("voxel", vox, rays, null, ReadEvents(Path.Combine(dir, "voxel.xev"), N), 1);
        bad += Run("voxel_excl", vox, rays, excl1, ReadEvents(Path.Combine(dir, "voxel_excl.xev"), N), 2);
        var oct = new Octree(model, OD, OP);
        bad += Run("octree", oct, rays, null, ReadEvents(Path.Combine(dir, "octree.xev"), N), 3);
        bad += Run("octree_excl", oct, rays, excl1, ReadEvents(Path.Combine(dir, "octree_excl.xev"), N), 4);
        var kd = new KDTree(model, KDD, KDP);
        bad += Run("kdtree", kd, rays, null, ReadEvents(Path.Combine(dir, "kdtree.xev"), N), 5);
#if HARE_GPU
        if (gpu)
        {   // the drop-in classes of this repo, single-ray (host path) and batch (HIP kernels), against the same records
            using (var g = new Gpu_Voxel_Grid(model, D))
            {
                bad += Run("gpu voxel one", g, rays, null, ReadEvents(Path.Combine(dir, "voxel.xev"), N), 6);
                bad += Run("gpu voxel excl", g, rays, excl1, ReadEvents(Path.Combine(dir, "voxel_excl.xev"), N), 7);
                var R = new Ray[N];
                for (int i = 0; i < N; i++) R[i] = new Ray(rays[6 * i], rays[6 * i + 1], rays[6 * i + 2], rays[6 * i + 3], rays[6 * i + 4], rays[6 * i + 5], 0, i + 1);
                var ev = new X_Event[N];
                g.Shoot(R, 0, ev);
                var got = new Rec[N];
                for (int i = 0; i < N; i++) got[i] = FromEvent(ev[i]);
                bad += Compare("gpu voxel batch", got, ReadEvents(Path.Combine(dir, "voxel.xev"), N), 0);
            }
            using (var g = new Gpu_Octree(model, OD, OP)) bad += Run("gpu octree one", g, rays, null, ReadEvents(Path.Combine(dir, "octree.xev"), N), 8);
            using (var g = new Gpu_KDTree(model, KDD, KDP)) bad += Run("gpu kdtree one", g, rays, null, ReadEvents(Path.Combine(dir, "kdtree.xev"), N), 9);
        }
#else
        if (gpu) Console.WriteLine("--gpu: rebuild with -p:DefineConstants=HARE_GPU and the shim sources (HareHip.cs, Gpu_Spatial_Partition.cs)");
#endif
        bad += Case2(Path.Combine(dir, "c2"));
        bad += Case3(Path.Combine(dir, "c3"));
        Console.WriteLine(bad == 0 ? "PINNED: the reference reproduces every committed X_Event bit for bit" : "MISMATCH: " + bad + " records differ");
        return bad == 0 ? 0 : 1;
    }
}
